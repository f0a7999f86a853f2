"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C ABI of
libpangene_amd.so; the oracle (host driver + plain-C backend) and the committed reference md5s are the
checkers."""
import ctypes as C
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, all_cases, golden_files
import oracle_host  # tests/oracle_host.py: the checker build of the host driver
from pangene_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip(built):
    lib = capi.load()
    C.c_int.in_dll(lib, "pg_verbose").value = 0
    return lib


@pytest.fixture(scope="module")
def ora(built):
    lib = oracle_host.load()
    C.c_int.in_dll(lib, "pg_verbose").value = 0
    return lib


def test_device_primitives(hip):
    raw = C.CDLL(capi.LIB_HIP)
    rng = np.random.default_rng(1)
    for n, nb in [(1, 8), (63, 16), (2048, 24), (2049, 24), (5000, 40), (1_000_003, 37)]:
        k = rng.integers(0, 1 << nb, size=n, dtype=np.uint64)
        k[: n // 3] &= np.uint64(0xFF)  # many ties: stability matters
        v = np.arange(n, dtype=np.uint32)
        k2, v2 = k.copy(), v.copy()
        assert raw.pga_selftest_sort(k2.ctypes.data_as(C.c_void_p), v2.ctypes.data_as(C.c_void_p), C.c_int64(n), C.c_int32(nb)) == 0
        o = np.argsort(k, kind="stable")
        assert np.array_equal(k2, k[o]) and np.array_equal(v2, v[o])
    for n in [1, 100, 1024, 1025, 300000, 2_000_001, 5_000_011]:  # the last one has more than 2048 tiles: the three-launch form of the scan
        a = rng.integers(-5, 50, size=n).astype(np.int32)
        seg = np.sort(rng.integers(0, max(1, n // 7), size=n)).astype(np.int32)
        out = np.zeros(n, dtype=np.int32)
        for mode in (0, 1, 2):
            assert raw.pga_selftest_scan(a.ctypes.data_as(C.c_void_p), seg.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                         C.c_int64(n), C.c_int32(mode)) == 0
            if mode == 0:
                exp = np.concatenate(([0], np.cumsum(a[:-1].astype(np.int64)))).astype(np.int32)
            elif mode == 1:
                exp = np.concatenate(([-1], np.maximum.accumulate(np.maximum(a, -1))[:-1])).astype(np.int32)
            else:
                # running maximum inside the segments: seg is sorted and a in [-5, 50), so a + 100 seg restarts above every earlier value
                exp = (np.maximum.accumulate(a.astype(np.int64) + 100 * seg.astype(np.int64)) - 100 * seg.astype(np.int64)).astype(np.int32)
            assert np.array_equal(out, exp), (n, mode)


@pytest.mark.parametrize("args", ["", "-p0 -a1", "-S", "-f0.3"])
def test_sweep_slow_list_and_list_overflow(hip, ora, tmp_path, args):
    """synth.dense: hits with partners hundreds of slots away and waves with > 512 overlapping pairs leave the LDS pair
    list for k_sweep_slow; the result must not depend on which path a hit took (HIP == oracle, which == the reference)"""
    fs = synth.write_files(synth.dense(3), str(tmp_path / "d"))
    for mode in (1, 2):
        hip.pg_set_exact_mode(mode); ora.pg_set_exact_mode(mode)
        a, b = capi.run(hip, fs, args.split()), capi.run(ora, fs, args.split())
        assert a == b and len(a) > 1000
    hip.pg_set_exact_mode(1); ora.pg_set_exact_mode(1)


@pytest.mark.parametrize("args", ["", "-p0 -a1", "-G"])
def test_vertex_fold_spills_beyond_eight_dominators(hip, ora, tmp_path, args):
    """synth.many_doms: genes with 11 distinct dominators over the genomes overflow the per-gene slots of k_vtx_fold"""
    fs = synth.write_files(synth.many_doms(2), str(tmp_path / "m"))
    a, b = capi.run(hip, fs, args.split()), capi.run(ora, fs, args.split())
    assert a == b and len(a) > 1000


@pytest.mark.parametrize("name,variant", [("C4", ""), ("C4", "-p0 -a1"), ("bact20", ""), ("human8f", "-S")])
def test_reference_main_c_on_the_product_library(built, expected, name, variant):
    """oracle/_ref/pangene_main_on_amd = the reference's unmodified main.c linked against libpangene_amd.so (built in the build
    container, oracle/Makefile): the drop-in claim as an executable, on the GPU"""
    exe = os.path.join(ROOT, "oracle", "_ref", "pangene_main_on_amd")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/pangene_main_on_amd was not built")
    env = dict(os.environ, PANGENE_EXACT="all")
    r = subprocess.run([exe] + variant.split() + golden_files(name), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    assert hashlib.md5(r.stdout).hexdigest() == expected[name][variant]["md5"]


_ENV_RUN = r'''
import sys, ctypes as C
sys.path.insert(0, %r)
from pangene_amd import capi
lib = capi.load(); C.c_int.in_dll(lib, "pg_verbose").value = 0
lib.pg_set_exact_mode(int(sys.argv[2]))
open(sys.argv[1], "wb").write(capi.run(lib, sys.argv[4:], sys.argv[3].split()))
''' % ROOT


def _run_with_env(tmp_path, env, mode, variant, files):
    outp = str(tmp_path / "o.bin")
    r = subprocess.run([sys.executable, "-c", _ENV_RUN, outp, str(mode), variant] + files, env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return open(outp, "rb").read()


@pytest.mark.parametrize("name,variant", [("bact20", ""), ("human8f", "-p0 -a1"), ("fuzz3", "-S"), ("dense", ""), ("manydoms", "-G"), ("human8", "--bed=flag"), ("fuzz7126", "-D 300 -C 2")])
@pytest.mark.parametrize("env", [{"PANGENE_ARC_SORT_PATH": "1", "PANGENE_WAIT": "sync"}, {"PANGENE_GENE_TABLE_LOG2": "2", "PANGENE_ROUND_FILTER_HOST": "1"}, {"PANGENE_VTX_SPILL_CAP": "3", "PANGENE_RANK_BY_SORT": "1", "PANGENE_PAIR_SCAN_GENERAL": "1", "PANGENE_LOOP": "nopre"},
                                 {"PANGENE_BRANCH_LOOP_HOST": "1", "PANGENE_GLOBAL_SORT": "1", "PANGENE_FILTERS": "global"}, {"PANGENE_LOOP": "nofinal", "PANGENE_MERGE_LITERAL": "1", "PANGENE_SWEEP_LISTS": "global"},
                                 {"PANGENE_FILTERS": "k32", "PANGENE_LOOP": "noskip", "PANGENE_SWEEP_LISTS": "lds"}, {"PANGENE_BIN_CAP": "256", "PANGENE_LIVE_LISTS": "2"}, {"PANGENE_BIN_CAP": "2048", "PANGENE_BINS": "1", "PANGENE_LOOP_FUSE": "0", "PANGENE_LOOP_PAIR_CAP": "8"}, {"PANGENE_LOOP_PAIR_CAP": "24", "PANGENE_POST": "lds"}, {"PANGENE_LIVE_LISTS": "1", "PANGENE_LIVE": "fullsweep", "PANGENE_RANK": "scan", "PANGENE_Y_FIXUP": "0", "PANGENE_FLAG_VTX": "x"}])
def test_arc_round_paths_agree(hip, expected, tmp_path, name, variant, env):
    """pg_gen_arc has two formulations on the device: the gene-major one (k_genes.hpp, the default) and the reference's global sort
    (the path of rounds in which a hub gene overflows the per-gene LDS table).  Forcing the sort path, and shrinking the table to 4
    entries so that most rounds overflow, must both reproduce the reference's bytes (mode all).  Third setting: a vertex spill area of
    three records (manydoms then needs the second, grown attempt of pga_vtx_partials) and the 64-bit comparison keys ranked by a sort
    (the path of shards whose score, preferred bit and protein rank do not fit 32 bits), and the general scan for the pair offsets
    (the path of graphs with more than 65536 oriented vertices), and graph 2 (graph.c:293-298) host-driven in front of the queued branch
    rounds instead of as their pre-step.  The second setting also keeps pg_flt_high_occ's tests on the host
    (the general route of a branch round: sharded runs, log lines, rounds repeated on the sort path) -- and with a four-entry table
    the queued branch rounds (pga_branch_loop, the default) give up on their sticky flag, so the run is repeated with host-driven
    rounds (RC_REDO).  Fourth setting: host-driven rounds from the start (one wait per round, verdicts of pg_flt_high_occ fetched as
    bytes: the default of round 2) and stage A's orders by the multi-workgroup radix sort instead of k_genome_sort (the path of
    genomes with more than 25 600 hits), and read.c:249-256 by the four kernels with their tables in HBM instead of k_genome_filters (the
    path of shards whose P + 8 Q bytes do not fit the LDS).  Fifth setting: the last branch round and the arc round of the graph that is written driven by the
    host behind the queued rounds (the default queues them too and renumbers segments and arcs on the host at the end); and every exon-list
    merge of the sweeps by the reference's literal steps (cds_inter_ref) instead of the shortcuts of cds_inter_t.  Sixth setting: the 4-byte `best`
    entries of k_genome_filters (the form of shards whose gene tables would not fit the LDS with 8-byte entries: 20 000-gene human
    annotations) on these small shards too, and every queued branch round run in full (no fixed-point gates).  Seventh and eighth setting:
    stage A's orders by CONTIG BINS (k_segsort.hpp: the planes grouped by contig at the upload, a workgroup per run of consecutive contigs -- the
    path of genomes beyond one workgroup's LDS, a human assembly) with bins of at most 256 / 2048 hits, so that these small genomes are cut into
    several bins or sorted as one bin out of the grouped planes; the seventh also asks in every queued round whether the live lists are worth
    building again (PANGENE_LIVE_LISTS=2).  Ninth setting: live lists, but the sweeps of the rounds over every record instead of the members' compact
    records (SweepView::xmap), the walkable ranks of pg_gen_rep_pos by the general scan instead of k_rank_genome, the cm order by radix passes
    instead of transpositions."""
    out = _run_with_env(tmp_path, env, 2, variant, golden_files(name))
    assert hashlib.md5(out).hexdigest() == expected[name][variant]["md5"]


def test_cross_shard_arc_merge(hip):
    """pga_arc_merge (what every rank runs on the all-gathered arc tables of a sharded round) against a numpy reduce-by-key"""
    raw = C.CDLL(capi.LIB_HIP)
    dt = np.dtype([("x", "<u8"), ("n_genome", "<i4"), ("tot_cnt", "<i4"), ("sum_dist", "<u8"), ("sum_s1", "<i8"), ("sum_s2", "<i8")])
    assert dt.itemsize == 40
    rng = np.random.default_rng(7)
    for W, n_key, frac in [(1, 50, 1.0), (2, 1000, 0.7), (3, 5000, 0.5), (8, 20000, 0.6), (4, 10, 0.0), (5, 3000, 0.05)]:
        universe = np.unique(rng.integers(0, 1 << 14, size=n_key, dtype=np.uint64) << np.uint64(32) | rng.integers(0, 1 << 14, size=n_key, dtype=np.uint64))
        lists = []
        for r in range(W):
            keys = universe[rng.random(universe.size) < frac]  # sorted, unique
            a = np.zeros(keys.size, dtype=dt)
            a["x"] = keys
            a["n_genome"] = rng.integers(1, 50, size=keys.size); a["tot_cnt"] = a["n_genome"] + rng.integers(0, 9, size=keys.size)
            a["sum_dist"] = rng.integers(0, 1 << 40, size=keys.size, dtype=np.uint64)
            a["sum_s1"] = rng.integers(0, 1 << 33, size=keys.size); a["sum_s2"] = rng.integers(0, 1 << 33, size=keys.size)
            lists.append(a)
        cnt = np.array([len(a) for a in lists], dtype=np.int64)
        slot = int(max(1, cnt.max()))
        g = np.zeros(W * slot, dtype=dt)
        for r, a in enumerate(lists):
            g[r * slot: r * slot + len(a)] = a
        out = np.zeros(int(cnt.sum()) + 1, dtype=dt)
        n_out = C.c_int64(0)
        assert raw.pga_selftest_merge(g.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p), C.c_int32(W), C.c_int64(slot),
                                      out.ctypes.data_as(C.c_void_p), C.byref(n_out)) == 0
        allv = np.concatenate(lists) if cnt.sum() else np.zeros(0, dtype=dt)
        ux, inv = np.unique(allv["x"], return_inverse=True)
        assert n_out.value == ux.size
        got = out[: ux.size]
        assert np.array_equal(got["x"], ux)
        for f in ("n_genome", "tot_cnt", "sum_dist", "sum_s1", "sum_s2"):
            exp = np.zeros(ux.size, dtype=np.int64)
            np.add.at(exp, inv, allv[f].astype(np.int64))
            assert np.array_equal(got[f].astype(np.int64), exp), (W, f)


@pytest.mark.parametrize("name,variant,mode", [("bact20", "", 1), ("bact20", "-D 600 -C 3 -F", 2), ("human8f", "-p0 -a1", 1), ("fuzz7115", "-D 1000 -C 1 -p0 -a1", 2), ("manydoms", "-G", 1)])
def test_repeated_runs_give_the_same_bytes(hip, expected, name, variant, mode):
    """Forty runs in one process, one answer.  This is the detector for host/device ordering mistakes: the host reads counters,
    degrees and n_dist_loci out of pinned memory after a wait, and a wait that does not release the device's writes at system scope
    (the doorbell kernel this library once used) shows up as a different GFA once in a few hundred runs, on some boxes only."""
    hip.pg_set_exact_mode(mode)
    files, args = golden_files(name), variant.split()
    first = capi.run(hip, files, args)
    if variant in expected[name] and "md5" in expected[name][variant]:
        assert hashlib.md5(first).hexdigest() == expected[name][variant]["md5"]
    for _ in range(39):
        assert capi.run(hip, files, args) == first


@pytest.mark.parametrize("name,variant", [("C4", ""), ("bact20", "-S"), ("human8f", "-p0 -a1"), ("fuzz7126", "-D 300 -C 2"), ("manydoms", "-G"), ("dense", "")])
def test_poisoned_allocations_change_nothing(hip, expected, tmp_path, name, variant):
    """PANGENE_POISON=1 fills every fresh device and pinned allocation with a pattern: a kernel or the host reading memory that
    nobody wrote then faults or changes the output instead of passing on whatever an earlier context left there."""
    for mode in (2, 1):
        out = _run_with_env(tmp_path, {"PANGENE_POISON": "1"}, mode, variant, golden_files(name))
        assert hashlib.md5(out).hexdigest() == expected[name][variant]["md5"]


@pytest.mark.parametrize("name,variant", all_cases())
def test_hip_equals_reference_md5(hip, expected, name, variant):
    """exact mode 'all': bytes equal to the untouched reference's, --bed line order included"""
    hip.pg_set_exact_mode(2)
    out = capi.run(hip, golden_files(name), variant.split())
    assert hashlib.md5(out).hexdigest() == expected[name][variant]["md5"]


@pytest.mark.parametrize("name,variant", all_cases())
def test_hip_default_mode_equals_reference(hip, expected, name, variant):
    """default mode (auto + hazard escalation): GFA bytes equal the reference's; --bed compared as a set of lines"""
    hip.pg_set_exact_mode(1)
    out = capi.run(hip, golden_files(name), variant.split())
    e = expected[name][variant]
    if "md5_sorted" in e:
        assert hashlib.md5(b"\n".join(sorted(out.split(b"\n")))).hexdigest() == e["md5_sorted"]
    else:
        assert hashlib.md5(out).hexdigest() == e["md5"]


@pytest.mark.parametrize("name,variant", all_cases())
@pytest.mark.parametrize("mode", [0, 1])
def test_hip_equals_oracle(hip, ora, name, variant, mode):
    """same canonical order on both sides (modes off / auto): identical bytes, hazards or not.  One exception: the LINE ORDER of
    --bed output in mode auto.  The device finds the dominator of a hit with an atomicMax over its winners in whatever order they
    arrive, and reports hazard H3 when a winner meets an equal key that was the maximum at that moment; the oracle scans the winners
    in array order.  Both report every tie of the final maximum, the device also ties among keys that do not end up as the maximum
    (set mut2: 37 events on 25 contigs against 29 on 21): it may put more contigs on the reference's exact order than the oracle --
    harmless for the graph (and identical in mode all), but the BED lines of those contigs come in another order."""
    hip.pg_set_exact_mode(mode), ora.pg_set_exact_mode(mode)
    a, b = capi.run(hip, golden_files(name), variant.split()), capi.run(ora, golden_files(name), variant.split())
    if mode == 1 and "--bed" in variant:
        a, b = sorted(a.split(b"\n")), sorted(b.split(b"\n"))
    assert a == b


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
@pytest.mark.parametrize("variant", ["", "-p0 -a1", "-f 0.3", "--bed=flag"])
def test_irregular_exon_lists_hip_equals_oracle(hip, ora, tmp_path, seed, variant):
    """synth.odd_exons: U / V introns shorter than 3 bp, zero-length introns and exons -- exon lists that are not sorted and disjoint
    (the reference itself aborts on most of them: overlap.c:135 asserts cov_short <= 1).  The sweep's merges then may not take their
    shortcuts (early exit, shared exons in one step, bisection to the first overlapping exon: k_sweep.hpp cds_inter_t): k_prepare
    notices and every merge takes the reference's steps one by one (cds_inter_ref); HIP and oracle must agree byte for byte."""
    files = synth.write_files(synth.odd_exons(seed), str(tmp_path))
    for mode in (0, 2):
        hip.pg_set_exact_mode(mode), ora.pg_set_exact_mode(mode)
        assert capi.run(hip, files, variant.split()) == capi.run(ora, files, variant.split())


@pytest.mark.parametrize("name,variant", [("human8f", "-p0 -a1"), ("human8", ""), ("human8f", "-f 0.2"), ("mut1", "-S")])
def test_literal_merge_on_regular_lists_changes_nothing(hip, expected, tmp_path, name, variant):
    """PANGENE_MERGE_LITERAL=1: the step-by-step merge of overlap.c:17-33 instead of cds_inter_t's shortcuts, on ordinary data: same bytes"""
    if variant not in expected[name]:
        pytest.skip("no such golden variant")
    out = _run_with_env(tmp_path, {"PANGENE_MERGE_LITERAL": "1"}, 2, variant, golden_files(name))
    assert hashlib.md5(out).hexdigest() == expected[name][variant]["md5"]


@pytest.mark.parametrize("lists", ["lds", "global"])
@pytest.mark.parametrize("name,variant", [("human8f", "-p0 -a1"), ("human8f", ""), ("human8", "-f 0.2"), ("mut1", "-S"), ("dense", ""), ("fuzz3", "-S"), ("manydoms", "-G")])
def test_sweep_exon_lists_in_lds_or_global_same_bytes(hip, expected, tmp_path, name, variant, lists):
    """K1 of stage A / pg_post_process comes in two builds of one body: k_sweep copies the tile's exon lists into LDS, k_sweep_lean reads them
    where they are (more workgroups on a CU); the host picks by the measured density of overlapping hits (k_list_density).
    PANGENE_SWEEP_LISTS fixes the choice: both give the reference's bytes on every case, whichever the density would have picked."""
    if variant not in expected[name]:
        pytest.skip("no such golden variant")
    out = _run_with_env(tmp_path, {"PANGENE_SWEEP_LISTS": lists}, 2, variant, golden_files(name))
    assert hashlib.md5(out).hexdigest() == expected[name][variant]["md5"]


@pytest.mark.parametrize("mode,live", [(1, "1"), (2, "1"), (2, "0")])
@pytest.mark.parametrize("name,variant", [("human8f", "-p0 -a1"), ("human8f", "-S"), ("human8", ""), ("bact20", ""), ("bact20", "-S"), ("mut1", "-S"), ("dense", ""), ("fuzz3", "-S"), ("fuzz7126", "-D 300 -C 2"),
                                          ("manydoms", "-G"), ("wide1", "-p0 -a1"), ("C4", ""), ("human8", "--bed=flag")])
def test_live_lists_same_bytes(hip, expected, tmp_path, name, variant, mode, live):
    """once few hits are left without flt the structures the rounds iterate over (the walk's cm-order list, the gene-major index,
    the half-arc records) hold those hits only (pga_ctx::live_on; by default when at most three hits in four are left at the vertex step).
    PANGENE_LIVE_LISTS=1 builds the lists whatever the share, =0 never: the reference's bytes either way -- in mode all every contig's order is
    replayed by the host, so every order override has to be told in the lists' coordinates as well (k_ovl_pos), with -S the index is built
    again after every cs override."""
    if variant not in expected[name]:
        pytest.skip("no such golden variant")
    out = _run_with_env(tmp_path, {"PANGENE_LIVE_LISTS": live}, mode, variant, golden_files(name))
    e = expected[name][variant]
    if mode == 1 and "md5_sorted" in e:  # (the line order of --bed in mode auto is the device's)
        assert hashlib.md5(b"\n".join(sorted(out.split(b"\n")))).hexdigest() == e["md5_sorted"]
    else:
        assert hashlib.md5(out).hexdigest() == e["md5"]


def _expected_large(name, variant):
    p = os.path.join(ROOT, "tests", "golden", "expected_large.json")
    if not os.path.exists(p):
        pytest.skip("tests/golden/expected_large.json missing (tests/golden/make_golden_large.py writes it in the build container)")
    import json
    e = json.load(open(p)).get(name, {}).get(variant)
    if e is None:
        pytest.skip("no reference md5 recorded for %s %r" % (name, variant))
    return e


def test_config1_full_size_against_reference(hip, tmp_path):
    """BASELINE configs[1]: bact(100, 5000), ~1 M hits: GFA bit-identical to the reference binary (run here when it was shipped,
    explicit skip of that half otherwise); rerun on the HBM-resident shard is idempotent."""
    files = synth.write_files_parallel("bact", str(tmp_path / "c2"), G=100, P=5000, seed=1)
    hip.pg_set_exact_mode(1)
    a = capi.run(hip, files, [])
    b = capi.run(hip, files, [])
    ref = os.path.join(ROOT, "oracle", "_ref", "pangene_ref")
    if not os.path.exists(ref):
        assert a == b
        pytest.skip("oracle/_ref/pangene_ref not shipped: the reference comparison did not run (idempotence did)")
    want = subprocess.run([ref] + files, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    assert a == want
    assert a == b
    s = sum(1 for l in a.split(b"\n") if l[:1] == b"S")
    assert 4000 < s <= 5000


@pytest.mark.parametrize("variant", ["", "-p0 -a1"])
def test_config2_human47_full_size_md5(hip, tmp_path_factory, variant):
    """BASELINE configs[2] stand-in at full size: 47 human-shaped haplotypes x 20 k multi-exon genes (fragmented contigs): the
    default tie-order mode must print the bytes whose md5 the untouched reference gave in the build container
    (tests/golden/expected_large.json).  This is the k_sweep<*, true> (exon records staged) flavour of K1 at size."""
    e = _expected_large("human47x20k", variant)
    d = tmp_path_factory.getbasetemp() / "human47"
    if not d.exists():
        synth.write_files_parallel("human", str(d), G=47, Q=20000, iso=1.0, seed=1, frag=True)
    files = sorted(str(d / f) for f in os.listdir(d))
    hip.pg_set_exact_mode(1)
    out = capi.run(hip, files, variant.split())
    assert len(out) == e["bytes"] and hashlib.md5(out).hexdigest() == e["md5"]


def test_config3_per_gpu_shard_full_size_md5(hip, tmp_path):
    """the per-GPU shard of BASELINE configs[3] (1250 x 5 k bacterial genomes, ~12 M hits, past the Infinity Cache): md5 of the GFA
    equal to the untouched reference's (recorded in the build container)"""
    e = _expected_large("bact1250x5k", "")
    files = synth.write_files_parallel("bact", str(tmp_path / "c3"), G=1250, P=5000, seed=1)
    hip.pg_set_exact_mode(1)
    out = capi.run(hip, files, [])
    assert len(out) == e["bytes"] and hashlib.md5(out).hexdigest() == e["md5"]


def test_config4_per_gpu_shard_full_size_md5(hip, tmp_path):
    """the per-GPU shard of BASELINE configs[4] (200 assemblies x ~110 k all-isoform proteins over 8 GPUs, -p0 -a1): 25 human-shaped
    haplotypes x 20 k genes x 5.5 isoforms (~110 k proteins, ~2.8 M hits, genomes of ~110 k hits: beyond k_genome_sort's LDS budget,
    so stage A's orders take the multi-workgroup radix sort): md5 of the GFA equal to the untouched reference's"""
    e = _expected_large("human25x20k_iso5.5", "-p0 -a1")
    files = synth.write_files_parallel("human", str(tmp_path / "c4"), G=25, Q=20000, iso=5.5, seed=1, frag=True)
    hip.pg_set_exact_mode(1)
    out = capi.run(hip, files, ["-p0", "-a1"])
    assert len(out) == e["bytes"] and hashlib.md5(out).hexdigest() == e["md5"]


def _need_room(path, disk_gb, ram_gb):
    import shutil
    free = shutil.disk_usage(str(path)).free / 2**30
    ram = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") / 2**30
    if free < disk_gb or ram < ram_gb:
        pytest.skip("a full-size configuration needs %d GB of scratch disk and %d GB of host memory; this box has %.0f / %.0f" % (disk_gb, ram_gb, free, ram))


def test_config3_full_size_md5(hip, tmp_path):
    """BASELINE configs[3] at its FULL stated size on ONE MI355X (it fits: < 2^30 hits, ~100 GB of HBM): 10 000 bacterial genomes x 5 000
    proteins, ~97 M hits.  md5 of the GFA (S, L and the 10 000 W-lines) equal to the untouched reference's, which took hours of one core in
    the build container (tests/golden/expected_large.json: reference_wall_s)."""
    import shutil
    e = _expected_large("bact10000x5k", "")
    _need_room(tmp_path, 24, 96)
    files = synth.write_files_parallel("bact", str(tmp_path / "c3full"), G=10000, P=5000, seed=1)
    hip.pg_set_exact_mode(1)
    try:
        out = capi.run(hip, files, [])
    finally:
        shutil.rmtree(str(tmp_path / "c3full"), ignore_errors=True)
    assert len(out) == e["bytes"] and hashlib.md5(out).hexdigest() == e["md5"]


def test_config4_full_size_md5(hip, tmp_path):
    """BASELINE configs[4] at its FULL stated size on ONE MI355X: 200 human-shaped assemblies x 20 k genes x 5.5 isoforms (~110 k
    proteins, ~22 M multi-exon hits, genomes of ~110 k hits each), -p0 -a1: md5 of the GFA equal to the untouched reference's"""
    import shutil
    e = _expected_large("human200x20k_iso5.5", "-p0 -a1")
    _need_room(tmp_path, 12, 64)
    files = synth.write_files_parallel("human", str(tmp_path / "c4full"), G=200, Q=20000, iso=5.5, seed=1, frag=True)
    hip.pg_set_exact_mode(1)
    try:
        out = capi.run(hip, files, ["-p0", "-a1"])
    finally:
        shutil.rmtree(str(tmp_path / "c4full"), ignore_errors=True)
    assert len(out) == e["bytes"] and hashlib.md5(out).hexdigest() == e["md5"]


def test_empty_and_degenerate_inputs(hip, ora, tmp_path):
    p = tmp_path / "e"
    p.mkdir()
    (p / "a.paf").write_text("")
    (p / "b.paf").write_text("g1\t100\t0\t100\t+\tc1\t1000\t10\t310\t300\t300\t0\tms:i:400\tcg:Z:100M\n")
    (p / "c.paf").write_text("g1\t100\t0\t100\t-\tc1\t1000\t10\t310\t300\t300\t0\tms:i:400\tcg:Z:100M\n"
                             "g2\t100\t0\t100\t+\tc1\t1000\t400\t700\t300\t300\t0\tms:i:410\tcg:Z:100M\n")
    files = [str(p / x) for x in ("a.paf", "b.paf", "c.paf")]
    for args in ([], ["-p0"], ["--bed=raw"]):
        assert capi.run(hip, files, args) == capi.run(ora, files, args)


def test_fresh_seed_fuzz_hip_vs_oracle(hip, ora, tmp_path):
    """HIP == oracle on seeds no fixture holds (tests/fuzz_hip_vs_oracle.py, a bounded slice: ~300 comparisons): fuzz, bacterial
    and mutated shapes (non-positive scores, duplicated alignments, strand flips, ungrouped lines) x 7 option variants x both
    tie-order modes.  The seed base follows the calendar hour, so every run of the suite covers other inputs; PG_FUZZ_SEED pins it
    (the base is in the assertion message: a failure can be replayed)."""
    import time
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fuzz_hip_vs_oracle as fz
    first = int(os.environ.get("PG_FUZZ_SEED", 20000 + (int(time.time()) // 3600) % 50000 * 7))
    variants = [[], ["-p0", "-a1"], ["-S"], ["-F"], ["-b", "0.2", "-B", "0.1", "-y", "0.3"], ["-D", "300", "-C", "2"], ["-S", "-D", "600", "-C", "3"]]
    msgs = []
    tot, bad = fz.sweep(hip, ora, first, 7, str(tmp_path), variants=variants, human=False, log=msgs.append)
    assert tot >= 280
    assert not bad, "first seed %d: %s" % (first, "; ".join(bad[:5]))


def test_exchange_aliases_device_memory():
    """the exchange hook must operate IN PLACE on library-owned HBM: torch.as_tensor on a raw pointer may not copy"""
    import torch
    from pangene_amd import exchange
    base = torch.arange(64, dtype=torch.int32, device="cuda")
    t = exchange._tensor(base.data_ptr(), 64 * 4, 1, base.device).view(torch.int32)
    t.add_(5)
    torch.cuda.synchronize()
    assert base[3].item() == 8 and t.data_ptr() == base.data_ptr()


def test_forced_exchange_single_rank_native_rccl(built, tmp_path):
    """one GPU, world size 1, every collective issued by the library itself through RCCL on its own stream: same GFA"""
    import sys
    files = synth.write_files(synth.bact(10, 300, seed=3), str(tmp_path / "x"))
    code = r'''
import sys, os, ctypes as C
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from pangene_amd import capi, exchange
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29534", RANK="0", WORLD_SIZE="1", PANGENE_FORCE_EXCHANGE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
lib = capi.load(); C.c_int.in_dll(lib, "pg_verbose").value = 0
assert exchange.install_native(lib), lib.pg_rccl_error()
out = capi.run(lib, sys.argv[2:], [])
open(sys.argv[1], 'wb').write(out)
lib.pg_rccl_finalize()
dist.destroy_process_group()
''' % ROOT
    outp = str(tmp_path / "x.gfa")
    r = subprocess.run([sys.executable, "-c", code, outp] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lib = capi.load()
    C.c_int.in_dll(lib, "pg_verbose").value = 0
    assert open(outp, "rb").read() == capi.run(lib, files, [])


def test_forced_exchange_single_rank_nccl(built, tmp_path):
    """one GPU, world size 1, every collective routed through torch.distributed/nccl (RCCL): same GFA"""
    import sys
    files = synth.write_files(synth.bact(10, 300, seed=3), str(tmp_path / "x"))
    code = r'''
import sys, os, ctypes as C
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from pangene_amd import capi, exchange
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", PANGENE_FORCE_EXCHANGE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
lib = capi.load(); C.c_int.in_dll(lib, "pg_verbose").value = 0
keep = exchange.install(lib, device=torch.device("cuda", 0))
out = capi.run(lib, sys.argv[2:], [])
open(sys.argv[1], 'wb').write(out)
dist.destroy_process_group()
''' % ROOT
    outp = str(tmp_path / "x.gfa")
    r = subprocess.run([sys.executable, "-c", code, outp] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lib = capi.load()
    C.c_int.in_dll(lib, "pg_verbose").value = 0
    assert open(outp, "rb").read() == capi.run(lib, files, [])


def _rank_on_shared_gpu(rank, world, port, files, variant, cuts, q, verbose=None):
    import torch, torch.distributed as dist
    sys.path.insert(0, ROOT)
    from pangene_amd import capi as capi2, exchange
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    torch.cuda.set_device(0)
    lib = capi2.load()
    C.c_int.in_dll(lib, "pg_verbose").value = verbose[rank] if verbose else 0
    keep = exchange.install(lib, device=torch.device("cuda", 0))
    n = len(files)
    out = capi2.run(lib, files, variant, scan_only=[not (cuts[rank] <= k < cuts[rank + 1]) for k in range(n)])
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()
    del keep


@pytest.mark.parametrize("name,variant,cuts", [("bact20", "", [0, 10, 20]), ("human8f", "-p0 -a1", [0, 3, 8]), ("bact20", "", [0, 7, 13, 20]), ("fuzz2", "-F", [0, 2, 2, 99]),
                                               ("bact20", "host-driven", [0, 10, 20]), ("human8", "", [0, 2, 5, 8]), ("C4", "", [0, 16, 99]),
                                               # EIGHT ranks (the node's width), one of them without a genome
                                               ("bact20", "", [0, 3, 6, 6, 9, 12, 15, 18, 20]), ("human8f", "", [0, 1, 2, 3, 3, 4, 5, 6, 8]),
                                               # ranks with different log levels: the routes (and with them the collectives) follow one level all ranks agree on
                                               ("bact20", "verbose 3 0 1", [0, 7, 13, 20]), ("human8f", "verbose 0 3", [0, 3, 8])])
def test_sharded_hip_ranks_on_one_gpu(built, expected, name, variant, cuts, monkeypatch, capfd):
    """The whole sharded HIP path with W > 1 (id scan, partial vectors, cross-shard arc merge on the device, n_local sums; the branch
    rounds queued on every rank with their two collectives per round in between -- each waits for the stream first here):
    several ranks share this box's one GPU and exchange through gloo with host staging (RCCL will not put two ranks on
    one device).  Their combined output must be the reference's single-process GFA."""
    import socket
    import torch.multiprocessing as mp
    verbose = None
    if variant == "host-driven":  # the branch rounds of the sharded run driven by the host, not queued (sharded pga_branch_loop)
        monkeypatch.setenv("PANGENE_SHARDED_LOOP_HOST", "1")
        variant = ""
    elif variant.startswith("verbose"):
        verbose = [int(x) for x in variant.split()[1:]]
        variant = ""
    files = golden_files(name)
    cuts = [min(c, len(files)) for c in cuts]
    world = len(cuts) - 1
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_on_shared_gpu, args=(r, world, port, files, variant.split(), cuts, q, verbose)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    capfd.readouterr()  # (the level-3 ranks' log)
    sl = [b"\n".join(l for l in res[r].split(b"\n") if l[:1] in (b"S", b"L")) for r in range(world)]
    assert all(x == sl[0] for x in sl)
    w = b"\n".join(l for r in range(world) for l in res[r].split(b"\n") if l[:1] == b"W")
    whole = sl[0] + b"\n" + w + b"\n"
    assert hashlib.md5(whole).hexdigest() == expected[name][variant]["md5"]


@pytest.mark.parametrize("shape,key,variant", [("bact", "bact1250x5k", ""), ("human", "human25x20k_iso5.5", "-p0 -a1")])
def test_sharded_hip_eight_ranks_at_size_on_one_gpu(built, tmp_path, shape, key, variant):
    """W = 8 (the node's width) AT SIZE: the per-GPU shard of BASELINE configs[3] (1 250 bacterial genomes, 12.1 M hits) and of
    configs[4] (25 isoform-rich assemblies, 2.7 M hits, -p0 -a1), cut into eight ranks that share this box's one GPU and exchange over
    gloo -- device-resident slot merges of tens of thousands of arcs, pair lists of hundreds of thousands of pairs, the learned
    capacities, eight contexts side by side in HBM.  S/L lines of all ranks agree; with every rank's W lines they are the bytes
    whose md5 the untouched reference gave for the whole set (tests/golden/expected_large.json)."""
    import socket
    import torch.multiprocessing as mp
    e = _expected_large(key, variant)
    if shape == "bact":
        files = synth.write_files_parallel("bact", str(tmp_path / "s"), G=1250, P=5000, seed=1)
    else:
        files = synth.write_files_parallel("human", str(tmp_path / "s"), G=25, Q=20000, iso=5.5, seed=1, frag=True)
    world = 8
    cuts = [len(files) * r // world for r in range(world + 1)]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_on_shared_gpu, args=(r, world, port, files, variant.split(), cuts, q, None)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=1200) for _ in procs)
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    sl = [b"\n".join(l for l in res[r].split(b"\n") if l[:1] in (b"S", b"L")) for r in range(world)]
    assert all(x == sl[0] for x in sl)
    w = b"\n".join(l for r in range(world) for l in res[r].split(b"\n") if l[:1] == b"W")
    whole = sl[0] + b"\n" + w + b"\n"
    assert len(whole) == e["bytes"] and hashlib.md5(whole).hexdigest() == e["md5"]


def _rank_native_rccl(rank, world, port, files, variant, cuts, q):
    import torch, torch.distributed as dist
    sys.path.insert(0, ROOT)
    from pangene_amd import capi as capi2, exchange
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", rank))
    lib = capi2.load()
    C.c_int.in_dll(lib, "pg_verbose").value = 0
    assert exchange.install_native(lib), lib.pg_rccl_error()
    n = len(files)
    out = capi2.run(lib, files, variant, scan_only=[not (cuts[rank] <= k < cuts[rank + 1]) for k in range(n)])
    q.put((rank, out))
    dist.barrier()
    lib.pg_rccl_finalize()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,variant,world", [("bact20", "", 2), ("human8f", "-p0 -a1", 2), ("bact20", "-S", 4)])
def test_ranks_over_native_rccl(built, expected, name, variant, world):
    """one process per GPU, collectives issued by the library itself through RCCL on the kernels' stream (xGMI between the devices):
    the ranks' S/L lines agree and, with the W lines of all ranks, give the reference's single-process GFA.  Needs `world` GPUs:
    skipped on a smaller box (the 1-GPU boxes of this pool)."""
    import socket
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs, this box has %d" % (world, torch.cuda.device_count()))
    files = golden_files(name)
    cuts = [len(files) * r // world for r in range(world + 1)]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_native_rccl, args=(r, world, port, files, variant.split(), cuts, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    sl = [b"\n".join(l for l in res[r].split(b"\n") if l[:1] in (b"S", b"L")) for r in range(world)]
    assert all(x == sl[0] for x in sl)
    w = b"\n".join(l for r in range(world) for l in res[r].split(b"\n") if l[:1] == b"W")
    assert hashlib.md5(sl[0] + b"\n" + w + b"\n").hexdigest() == expected[name][variant]["md5"]


@pytest.mark.gpu
def test_pga_create_checks_abi_version_and_block_contents(hip):
    """A direct user of include/pangene_hip.h (not the host driver, which checks while it packs): pga_create refuses a shard built
    against another PGA_ABI_VERSION (PGA_ERR_ARG) and a block whose hits break the block's own declaration -- contig id >= n_ctg,
    cs beyond max_cs, exon range outside the exon list -- with PGA_ERR_RANGE, instead of indexing out of bounds later."""
    hdr = open(os.path.join(ROOT, "include", "pangene_hip.h")).read()
    import re
    abi = int(re.search(r"#define PGA_ABI_VERSION (\d+)u", hdr).group(1))

    class Block(C.Structure):
        _fields_ = [(k, C.c_int32) for k in ("n_hit", "n_exon", "n_ctg", "max_cs", "max_cm", "max_score_adj", "any_neg", "any_multi")] + [("data", C.c_void_p), ("n_words", C.c_size_t), ("vfirst", C.c_void_p), ("vbase", C.c_void_p)]

    class Shard(C.Structure):
        _fields_ = [("abi_version", C.c_uint32), ("n_genome", C.c_int32), ("n_genome_global", C.c_int32), ("genome_global", C.c_void_p), ("n_prot", C.c_int32),
                    ("n_gene", C.c_int32), ("n_hit", C.c_int64), ("n_exon", C.c_int64), ("block", C.c_void_p), ("prot_gid", C.c_void_p), ("gene_pref", C.c_void_p)]

    class Par(C.Structure):
        _fields_ = [("min_ov_ratio", C.c_double), ("check_strand", C.c_int32), ("drop_sgl_exon", C.c_int32), ("reserved", C.c_int32 * 4)]

    def attempt(version=abi, cid=0, cs=100, offx=0, vfirst=None, vbase=None):
        n, ne = 2, 2
        planes = np.zeros((10, n), np.int32)  # pid, contig, rank, score_ori, score_adj, n_exon, off_exon, cs, ce, cm
        planes[0] = [0, 1]; planes[1] = [0, cid]; planes[3] = planes[4] = 50; planes[5] = 1; planes[6] = [0, offx]
        planes[7] = [10, cs]; planes[8] = planes[7] + 90; planes[9] = planes[7] + 45
        words = np.concatenate([planes.ravel(), np.zeros(1, np.int32), np.array([0, 90, 0, 90], np.int32)])
        vf = np.asarray(vfirst, np.int32) if vfirst is not None else None
        vb = np.asarray(vbase, np.int64) if vbase is not None else None
        blk = Block(n, ne, 1 if vf is None else len(vf), 100, 145, 50, 0, 0, words.ctypes.data, words.size, vf.ctypes.data if vf is not None else None, vb.ctypes.data if vb is not None else None)
        gg, pg, pref = np.zeros(1, np.int32), np.array([0, 1], np.int32), np.zeros(2, np.uint8)
        sh = Shard(version, 1, 1, gg.ctypes.data, 2, 2, n, ne, C.addressof(blk), pg.ctypes.data, pref.ctypes.data)
        par, ctx = Par(0.5, 0, 0), C.c_void_p()
        hip.pga_create.restype = C.c_int
        rc = hip.pga_create(C.byref(ctx), C.byref(sh), C.byref(par))
        if ctx.value:
            hip.pga_destroy.restype = None
            hip.pga_destroy(ctx)
        return rc

    assert attempt() == 0
    assert attempt(version=abi + 1) == -3  # PGA_ERR_ARG
    assert attempt(cid=1) == -2            # PGA_ERR_RANGE: contig 1 of a genome with one contig
    assert attempt(cs=101) == -2           # beyond the declared max_cs: the sort key would lose its top bit
    assert attempt(offx=2) == -2           # exon range outside the genome's exon list
    # virtual contigs (64-bit coordinates): both tables or none, pieces of a contig consecutive and in coordinate order
    assert attempt(cid=1, vfirst=[0, 0], vbase=[0, 1 << 40]) == 0
    assert attempt(cid=1, vfirst=[0, 0]) == -3
    assert attempt(cid=1, vfirst=[0, 0], vbase=[1 << 40, 0]) == -3   # bases go backwards
    assert attempt(cid=1, vfirst=[0, 2], vbase=[0, 0]) == -3         # a first piece that comes later
    assert attempt(cid=1, vfirst=[0, 1], vbase=[0, -1]) == -3


_XLOOP_CODE = r'''
import sys, os, ctypes as C, hashlib, json
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from pangene_amd import capi, exchange
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1], RANK="0", WORLD_SIZE="1", PANGENE_FORCE_EXCHANGE="1", PANGENE_TIMING="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
lib = capi.load(); C.c_int.in_dll(lib, "pg_verbose").value = 0
if sys.argv[2] == "native":
    assert exchange.install_native(lib), lib.pg_rccl_error()
else:
    keep = exchange.install(lib, device=torch.device("cuda", 0))
jobs = json.load(open(sys.argv[3]))
res = []
import tempfile
for files, variant, repeat in jobs:  # `repeat` passes over ONE resident shard (what bench.py's steps are): the context lives on
    sys.stderr.write("JOB %%s %%r\n" %% (os.path.basename(os.path.dirname(files[0])), variant)); sys.stderr.flush()
    opt = capi.parse_args(lib, variant)
    d = lib.pg_data_init()
    capi.read_files(lib, opt, d, files)
    for k in range(repeat):
        if k and lib.pg_rerun_resident(d) != 0:
            raise RuntimeError("pg_rerun_resident failed")
        lib.pg_post_process(C.byref(opt), d)
        g = lib.pg_graph_init(d)
        lib.pg_graph_gen(C.byref(opt), g)
        if lib.pg_last_error():
            raise RuntimeError(lib.pg_last_error_str().decode())
        out = tempfile.mktemp(prefix="pangene_xloop_", suffix=".gfa")
        lib.pg_set_output(out.encode()); lib.pg_write_graph(g); lib.pg_write_walk(g); lib.pg_set_output(None)
        res.append(hashlib.md5(open(out, "rb").read()).hexdigest()); os.unlink(out)
        lib.pg_graph_destroy(g)
    lib.pg_data_destroy(d)
json.dump(res, open(sys.argv[4], "w"))
if sys.argv[2] == "native":
    lib.pg_rccl_finalize()
dist.destroy_process_group()
''' % ROOT


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["native", "torch"])
def test_sharded_branch_loop_forced_exchange(built, expected, tmp_path, kind):
    """The sharded form of pga_branch_loop (per round one all-gather of the ranks' slots + the merge on the device, one all-reduce of
    the n_local counts; no wait until the end) with world size 1 and every collective issued -- by the library itself through RCCL on
    the kernels' stream ("native": nothing waits), or through torch.distributed callbacks ("torch": each collective waits for the
    stream first).  The GFA is the reference's, and the log shows that the rounds really were queued (status 0), not host-driven."""
    import json
    import socket
    cases = [("bact20", ""), ("bact20", "-S"), ("C4", ""), ("human8", ""), ("human8f", "-p0 -a1"), ("mut1", ""), ("fuzz0", "-F")]
    cases = [(n, v) for n, v in cases if n in expected and v in expected[n]]
    jobs = [[golden_files(n), v.split(), 1] for n, v in cases]
    (tmp_path / "jobs.json").write_text(json.dumps(jobs))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-c", _XLOOP_CODE, str(port), kind, str(tmp_path / "jobs.json"), str(tmp_path / "res.json")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    err = r.stderr.decode()
    assert r.returncode == 0, err[-3000:]
    res = json.load(open(tmp_path / "res.json"))
    assert res == [expected[n][v]["md5"] for n, v in cases]
    assert err.count("rounds queued (sharded): backend status 0") >= len(cases) - 2, err[-3000:]  # (a data set may leave its rounds to the host: exact-order replay, a hub gene)


@pytest.mark.gpu
def test_sharded_branch_loop_learns_its_capacities(built, expected, tmp_path):
    """Exchange buffers that are too small (forced here) void the queued rounds: status 3, the run is repeated host-driven with the
    reference's result, and the next run over the same shard queues its rounds again with the capacities the failed one measured."""
    import json
    import socket
    jobs = [[golden_files("bact20"), [], 3]]
    (tmp_path / "jobs.json").write_text(json.dumps(jobs))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    for envx in ({"PANGENE_XLOOP_CAP": "64"}, {"PANGENE_XLOOP_CAP": "0,100"}):
        r = subprocess.run([sys.executable, "-c", _XLOOP_CODE, str(port), "native", str(tmp_path / "jobs.json"), str(tmp_path / "res.json")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600,
                           env=dict(os.environ, **envx))
        err = r.stderr.decode()
        assert r.returncode == 0, err[-3000:]
        assert json.load(open(tmp_path / "res.json")) == [expected["bact20"][""]["md5"]] * 3
        assert "rounds queued (sharded): backend status 3" in err and "rounds queued (sharded): backend status 0" in err, err[-3000:]
