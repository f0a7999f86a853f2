"""ctypes bindings of the C ABI in include/pangene_amd.h (the pangene.h-compatible surface).

Nothing here computes: every call goes into libpangene_amd.so (HIP backend).  (The checker build of the same host
driver -- linked against the plain-C oracle -- is loaded by tests/oracle_host.py, which hands its own path to load();
nothing in this package knows where it is.)
"""
from __future__ import annotations

import ctypes as C
import os
import shlex
import tempfile
from typing import List, Sequence

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_HIP = os.path.join(ROOT, "pangene_amd", "lib", "libpangene_amd.so")

PG_F_WRITE_BED_RAW, PG_F_WRITE_BED_WALK, PG_F_WRITE_BED_FLAG, PG_F_WRITE_NO_WALK = 0x1, 0x2, 0x4, 0x8
PG_F_WRITE_VTX_SEL, PG_F_FRAG_MODE, PG_F_NO_JOINT_PSEUDO, PG_F_ORI_FOR_BRANCH = 0x10, 0x20, 0x40, 0x80
PG_F_CHECK_STRAND, PG_F_DROP_SGL_EXON = 0x100, 0x200


class pg_opt_t(C.Structure):  # layout of pangene.h:23-42 (128 bytes)
    _fields_ = [("flag", C.c_uint32), ("gene_delim", C.c_int32), ("min_prot_ratio", C.c_double), ("min_prot_iden", C.c_double),
                ("score_adj_coef", C.c_double), ("min_ov_ratio", C.c_double), ("min_vertex_ratio", C.c_double),
                ("branch_diff", C.c_double), ("branch_diff_dist", C.c_double), ("branch_diff_cut", C.c_double),
                ("max_avg_occ", C.c_int32), ("max_degree", C.c_int32), ("max_dist_loci", C.c_int32), ("n_branch_flt", C.c_int32),
                ("min_arc_cnt", C.c_int32), ("local_dist", C.c_int32), ("local_count", C.c_int32),
                ("excl", C.c_void_p), ("incl", C.c_void_p), ("preferred", C.c_void_p)]


ALLREDUCE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32)
ALLGATHER_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32)


class pg_exchange_t(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("user", C.c_void_p), ("allreduce", ALLREDUCE_CB), ("allgather", ALLGATHER_CB), ("stream_ordered", C.c_int32)]


_API = {
    "pg_opt_init": (None, [C.POINTER(pg_opt_t)]),
    "pg_data_init": (C.c_void_p, []),
    "pg_data_destroy": (None, [C.c_void_p]),
    "pg_read_paf": (C.c_int32, [C.POINTER(pg_opt_t), C.c_void_p, C.c_char_p]),
    "pg_scan_paf_ids": (C.c_int32, [C.POINTER(pg_opt_t), C.c_void_p, C.c_char_p]),
    "pg_read_paf_batch": (C.c_int32, [C.POINTER(pg_opt_t), C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint8), C.c_int32]),
    "pg_post_process": (None, [C.POINTER(pg_opt_t), C.c_void_p]),
    "pg_graph_init": (C.c_void_p, [C.c_void_p]),
    "pg_graph_gen": (None, [C.POINTER(pg_opt_t), C.c_void_p]),
    "pg_graph_destroy": (None, [C.c_void_p]),
    "pg_write_bed": (None, [C.c_void_p, C.c_int32]),
    "pg_write_graph": (None, [C.c_void_p]),
    "pg_write_walk": (None, [C.c_void_p]),
    "pg_write_matrix": (None, [C.c_void_p, C.c_int32]),
    "pg_gfa2matrix_file": (C.c_int, [C.c_char_p, C.c_int32, C.c_char_p, C.c_int32]),
    "pg_read_list_dict": (C.c_void_p, [C.c_char_p]),
    "pg_dict_destroy": (None, [C.c_void_p]),
    "pg_last_error": (C.c_int, []),
    "pg_last_error_str": (C.c_char_p, []),
    "pg_set_output": (C.c_int, [C.c_char_p]),
    "pg_set_exchange": (None, [C.POINTER(pg_exchange_t)]),
    "pg_last_path_seconds": (C.c_double, []),
    "pg_last_path_hits": (C.c_int64, []),
    "pg_last_attempts": (C.c_int, []),
    "pg_device_copy_gbps": (C.c_double, [C.c_size_t, C.c_int32]),
    "pg_trim_host_cache": (None, [C.c_size_t]),
    "pg_last_upload_seconds": (C.c_double, []),
    "pg_last_pack_seconds": (C.c_double, []),
    "pg_last_reserve_seconds": (C.c_double, []),
    "pg_shard_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pg_rerun_resident": (C.c_int, [C.c_void_p]),
    "pg_kernel_timing": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pg_kernel_timing_reset": (C.c_int, [C.c_void_p]),
    "pg_collective_count": (C.c_int64, []),
    "pg_set_exact_mode": (None, [C.c_int]),
    "pg_phase_times": (C.c_int, [C.POINTER(C.c_double), C.c_int]),
    "pg_phase_name": (C.c_char_p, [C.c_int]),
}

# only in the HIP product library (the oracle-host test library has no RCCL exchange)
_API_HIP_ONLY = {
    "pg_rccl_unique_id": (C.c_int, [C.c_void_p]),
    "pg_rccl_init": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p]),
    "pg_rccl_finalize": (C.c_int, []),
    "pg_rccl_error": (C.c_char_p, []),
}


def load(path: str = LIB_HIP) -> C.CDLL:
    """The product library (default), or another build of the pangene.h surface at `path` (tests: the checker build)."""
    hip = os.path.abspath(path) == os.path.abspath(LIB_HIP)
    if not os.path.exists(path):
        raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` first" % path)
    lib = C.CDLL(path, mode=os.RTLD_LOCAL | os.RTLD_NOW)
    for name, (res, args) in _API.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if hip:
        for name, (res, args) in _API_HIP_ONLY.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    return lib


def parse_args(lib: C.CDLL, argv: Sequence[str]) -> pg_opt_t:
    """The option letters of the reference's main.c:70-113."""
    opt = pg_opt_t()
    lib.pg_opt_init(C.byref(opt))
    it = iter(argv)
    for a in it:
        if a == "-J": opt.flag |= PG_F_NO_JOINT_PSEUDO
        elif a == "-E": opt.flag |= PG_F_DROP_SGL_EXON
        elif a == "-F": opt.flag |= PG_F_FRAG_MODE
        elif a == "-S": opt.flag |= PG_F_CHECK_STRAND
        elif a == "-w": opt.flag |= PG_F_WRITE_NO_WALK
        elif a == "-G": opt.flag |= PG_F_WRITE_VTX_SEL
        elif a == "--ori-sc": opt.flag |= PG_F_ORI_FOR_BRANCH
        elif a in ("--bed", "--bed=walk"): opt.flag |= PG_F_WRITE_BED_WALK
        elif a == "--bed=raw": opt.flag |= PG_F_WRITE_BED_RAW
        elif a == "--bed=flag": opt.flag |= PG_F_WRITE_BED_FLAG
        elif a in ("--matrix", "--matrix=presence", "--matrix=count"): pass  # handled by run()
        elif a[:2] in ("-p", "-a", "-f", "-c", "-g", "-r", "-b", "-B", "-y", "-T", "-D", "-C", "-e", "-l", "-m", "-d", "-X", "-I", "-P"):
            v = a[2:] if len(a) > 2 else next(it)
            k = a[1]
            if k == "p": opt.min_vertex_ratio = float(v)
            elif k == "a": opt.min_arc_cnt = int(v)
            elif k == "f": opt.min_ov_ratio = float(v)
            elif k == "c": opt.max_avg_occ = int(v)
            elif k == "g": opt.max_degree = int(v)
            elif k == "r": opt.max_dist_loci = int(v)
            elif k == "b": opt.branch_diff = float(v)
            elif k == "B": opt.branch_diff_cut = float(v)
            elif k == "y": opt.branch_diff_dist = float(v)
            elif k == "T": opt.n_branch_flt = int(float(v))
            elif k == "D": opt.local_dist = int(float(v) + .499)
            elif k == "C": opt.local_count = int(v)
            elif k == "e": opt.min_prot_iden = float(v)
            elif k == "l": opt.min_prot_ratio = float(v)
            elif k == "m": opt.score_adj_coef = float(v)
            elif k == "d": opt.gene_delim = ord(v[0])
            elif k == "X": opt.excl = lib.pg_read_list_dict(v.encode())
            elif k == "I": opt.incl = lib.pg_read_list_dict(v.encode())
            elif k == "P": opt.preferred = lib.pg_read_list_dict(v.encode())
        else:
            raise ValueError("unknown option " + a)
    return opt


def read_files(lib: C.CDLL, opt, d, files: Sequence[str], scan_only: Sequence[bool] | None = None, n_threads: int = 0, batch: bool = True) -> int:
    """main.c:121-122 for all files: parsed on host threads (pg_read_paf_batch) or one pg_read_paf / pg_scan_paf_ids per file."""
    n = len(files)
    if not batch:
        rc = 0
        for k, f in enumerate(files):
            fn = lib.pg_scan_paf_ids if scan_only is not None and scan_only[k] else lib.pg_read_paf
            rc += min(0, fn(C.byref(opt), d, f.encode()))
        return rc
    fns = (C.c_char_p * max(n, 1))(*[f.encode() for f in files])
    mask = (C.c_uint8 * max(n, 1))(*[1 if scan_only is not None and scan_only[k] else 0 for k in range(n)])
    return lib.pg_read_paf_batch(C.byref(opt), d, n, fns, mask, n_threads)


def run(lib: C.CDLL, files: Sequence[str], argv: Sequence[str] = (), scan_only: Sequence[bool] | None = None, batch: bool = True) -> bytes:
    """main.c:117-142 in-process: returns what the command line would print to stdout."""
    opt = parse_args(lib, argv)
    fd, out = tempfile.mkstemp(prefix="pangene_", suffix=".out")
    os.close(fd)
    lib.pg_set_output(out.encode())
    d = lib.pg_data_init()
    try:
        read_files(lib, opt, d, files, scan_only, batch=batch)
        lib.pg_post_process(C.byref(opt), d)
        if lib.pg_last_error():
            raise RuntimeError("pangene_amd: " + lib.pg_last_error_str().decode())
        if opt.flag & PG_F_WRITE_BED_RAW:
            lib.pg_write_bed(d, 0)
        else:
            g = lib.pg_graph_init(d)
            lib.pg_graph_gen(C.byref(opt), g)
            if lib.pg_last_error():
                raise RuntimeError("pangene_amd: " + lib.pg_last_error_str().decode())
            if any(x.startswith("--matrix") for x in argv): lib.pg_write_matrix(g, 1 if "--matrix=count" in argv else 0)
            elif opt.flag & PG_F_WRITE_BED_WALK: lib.pg_write_bed(d, 1)
            elif opt.flag & PG_F_WRITE_BED_FLAG: lib.pg_write_bed(d, 0)
            else:
                lib.pg_write_graph(g)
                if not (opt.flag & PG_F_WRITE_NO_WALK): lib.pg_write_walk(g)
            lib.pg_graph_destroy(g)
    finally:
        lib.pg_data_destroy(d)
        for h in (opt.excl, opt.incl, opt.preferred):
            if h: lib.pg_dict_destroy(h)
        lib.pg_set_output(None)
    with open(out, "rb") as fh:
        data = fh.read()
    os.unlink(out)
    return data
