"""64-bit contig coordinates (pangene.h:71: int64_t cs, cm, ce; read.c:202-204 parses them with strtol) through the 32-bit device
layout: the packer cuts long contigs into VIRTUAL contigs (include/pangene_hip.h, pga_genome_block_t; graph_driver.cpp:
virtual_contigs) and the backend puts them together again where the reference looks across hits (graph.c:113-121, branch.c:6-46).

The wide0..wide3 fixtures (tests/golden/make_golden.py: synth.widen, expected output from the untouched reference) already go
through every parity test of the suite.  Here: PANGENE_VCTG_PIECE shrinks the piece size so that the ORDINARY fixtures are cut into
hundreds of pieces too -- their output must not change by a byte -- and the per-contig gene matrix still adds up per contig."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

_CODE = r'''
import sys, os, ctypes as C, hashlib, json, io, gzip
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "oracle")); sys.path.insert(0, os.path.join(%r, "tests"))
from conftest import golden_files, GOLD
from pangene_amd import capi
import gfa2matrix_ref as ref, oracle_host
lib = (oracle_host.load() if sys.argv[1] == "oracle" else capi.load())
C.c_int.in_dll(lib, "pg_verbose").value = 0
exp = json.load(open(os.path.join(GOLD, "expected.json")))
bad = []
for name, variants in json.loads(sys.argv[2]):
    for v in variants:
        out = capi.run(lib, golden_files(name), v.split())
        e = exp[name][v]
        ok = hashlib.md5(b"\n".join(sorted(out.split(b"\n")))).hexdigest() == e["md5_sorted"] if "md5_sorted" in e else hashlib.md5(out).hexdigest() == e["md5"]
        if not ok: bad.append((name, v))
    gfa = capi.run(lib, golden_files(name), []).decode().split("\n")
    for cn in (0, 1):
        got = capi.run(lib, golden_files(name), ["--matrix=count"] if cn else ["--matrix"]).decode()
        if got != ref.gfa2matrix(gfa, copy_number=cn): bad.append((name, "--matrix", cn))
print(json.dumps(bad))
''' % (ROOT, ROOT, ROOT)

VARIANTS = ["", "-S", "-F", "-p0 -a1", "-D 600 -C 3 -F", "-S -D 600 -C 3", "-D 1000 -C 1 -p0 -a1", "--bed=walk", "--bed=flag", "-a2 -E", "-w"]
SETS = ["C4", "bact20", "human8", "human8f", "fuzz0", "fuzz4", "fuzz7115h", "mut1", "mut2", "dense", "wide0", "wide3"]


def _forced(backend, piece, sets):
    env = dict(os.environ, PANGENE_VCTG_PIECE=str(piece))
    jobs = [[s, VARIANTS] for s in sets]
    r = subprocess.run([sys.executable, "-c", _CODE, backend, json.dumps(jobs)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=1800)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert json.loads(r.stdout.decode().strip().split("\n")[-1]) == []


@pytest.mark.parametrize("piece", [1, 3000, 200000])
def test_forced_cuts_leave_every_output_unchanged(built, piece):
    """host driver + plain-C oracle backend; piece = 1: a cut at every hit-free gap"""
    _forced("oracle", piece, SETS)


@pytest.mark.gpu
@pytest.mark.parametrize("piece", [1, 200000])
def test_forced_cuts_on_the_gpu(built, piece):
    """the HIP backend: k_pack_yrec (contig identity and low words of 64-bit cm for the walk), the wide form of the (gene, genome)
    position records (k_rep_fill / k_n_local), hazard lists in terms of pieces"""
    _forced("hip", piece, SETS)
