"""`pangene --gpus N` (main.c:117-142 for N devices of a node): one command forks N - 1 workers, shards the PAF files, exchanges over
RCCL (HIP backend) or a shared-memory region (backends without a device), rank 0 prints the graph and gathers the ranks' lines.
CPU half: the command built on the oracle host (tests/_build/pangene_oraclehost) with 2, 3 and more processes than files must print
the reference's bytes.  GPU half: the product command on >= 2 devices (skipped below)."""
import hashlib
import os
import subprocess

import pytest

from conftest import ROOT, golden_files

ORA = os.path.join(ROOT, "tests", "_build", "pangene_oraclehost")
HIP = os.path.join(ROOT, "pangene_amd", "bin", "pangene")


def _run(exe, n, files, variant, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([exe, "--gpus", str(n)] + variant.split() + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return r.stdout


@pytest.mark.parametrize("n", [2, 3, 40])
@pytest.mark.parametrize("name,variant", [("bact20", ""), ("bact20", "-p0 -a1"), ("human8f", ""), ("human8f", "-S -D 600 -C 3"), ("fuzz0", "-F"), ("mut1", ""), ("C4", "-w"),
                                          ("manydoms", "-G"), ("human8", "--bed=walk"), ("fuzz3", "--bed=flag"), ("bact20", "--bed=raw"),
                                          ("wide0", ""), ("wide3", "-S"), ("wide1", "-D 300 -C 2")])  # (64-bit coordinates: virtual contigs on every rank)
def test_sharded_command_on_the_oracle_host(built, expected, name, variant, n):
    """2, 3 and 40 processes (more than files: empty ranks), shared-memory exchange: the reference's bytes (mode all pins the line
    order of the BED outputs too)"""
    out = _run(ORA, n, golden_files(name), variant, {"PANGENE_EXACT": "all"})
    assert hashlib.md5(out).hexdigest() == expected[name][variant]["md5"]


def test_sharded_command_default_mode_equals_one_process(built):
    files = golden_files("human8f")
    one = subprocess.run([ORA] + files, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    assert _run(ORA, 3, files, "") == one


def test_sharded_command_refuses_what_it_cannot_do(built):
    r = subprocess.run([ORA, "--gpus", "2", "--matrix"] + golden_files("C4"), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"--matrix" in r.stderr


@pytest.mark.parametrize("bad", [0, 1, 2])
def test_a_rank_that_fails_alone_takes_the_command_down(built, tmp_path, bad):
    """A rank that ends with an error after the go/no-go handshake used to leave the others waiting in a collective for ever (and the
    temporary files behind).  Rank 0 watches its workers, workers die with rank 0: the command ends, with an error, and cleans up."""
    e = dict(os.environ, PANGENE_FAULT_RANK=str(bad), TMPDIR=str(tmp_path))
    r = subprocess.run([ORA, "--gpus", "3"] + golden_files("bact20"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=60)
    assert r.returncode != 0
    assert b"injected fault" in r.stderr
    assert os.listdir(str(tmp_path)) == []


def test_ranks_with_different_log_levels_take_the_same_routes(built, expected):
    """-v3 on the command line: rank 0 logs at level 3, the workers are silenced to 1 -- the ROUTES (which pick the collectives) follow
    one level all ranks agree on, so the run ends and prints the reference's bytes"""
    out = _run(ORA, 3, golden_files("human8f"), "-v3", {"PANGENE_EXACT": "all"})
    assert hashlib.md5(out).hexdigest() == expected["human8f"][""]["md5"]


def test_files_are_cut_by_size_not_by_count(built, tmp_path):
    """contiguous blocks of genomes balanced by hit count -- one big file and many small ones"""
    files = golden_files("bact20")
    big = str(tmp_path / "big.paf")
    with open(big, "wb") as f:
        for fn in files[:10]:
            import gzip
            f.write(gzip.open(fn).read() if fn.endswith(".gz") else open(fn, "rb").read())
    one = subprocess.run([ORA, big] + files[10:], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    r = subprocess.run([ORA, "--gpus", "2", "-v3", big] + files[10:], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    assert r.stdout == one
    kept = [l for l in r.stderr.decode().split("\n") if "lines parsed" in l and "ids only" not in l]
    assert len(kept) == 1 and "big" in kept[0]  # rank 0's log: its shard is the big file alone


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 4])
@pytest.mark.parametrize("name,variant", [("bact20", ""), ("human8f", "-p0 -a1"), ("fuzz7126", "-D 300 -C 2"), ("human8", "--bed=walk"), ("wide0", "")])
def test_sharded_command_on_gpus(built, expected, name, variant, n):
    """the product command over native RCCL, one process per device; needs n devices"""
    import torch
    if torch.cuda.device_count() < n:
        pytest.skip("needs %d GPUs, this box has %d" % (n, torch.cuda.device_count()))
    out = _run(HIP, n, golden_files(name), variant, {"PANGENE_EXACT": "all"})
    assert hashlib.md5(out).hexdigest() == expected[name][variant]["md5"]


@pytest.mark.gpu
def test_command_on_one_gpu(built, expected):
    """the product command itself (a fresh process: HIP initialisation, code-object load, parsing, path, GFA text)"""
    r = subprocess.run([HIP] + golden_files("bact20"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-1000:]
    assert hashlib.md5(r.stdout).hexdigest() == expected["bact20"][""]["md5"]
