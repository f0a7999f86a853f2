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
                                          ("manydoms", "-G"), ("human8", "--bed=walk"), ("fuzz3", "--bed=flag"), ("bact20", "--bed=raw")])
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


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 4])
@pytest.mark.parametrize("name,variant", [("bact20", ""), ("human8f", "-p0 -a1"), ("fuzz7126", "-D 300 -C 2"), ("human8", "--bed=walk")])
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
