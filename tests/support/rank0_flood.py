"""Forty thousand rank-0 hits of ONE protein in one stretch of hits (the ranks edited in place between pg_read_paf and pg_post_process, as the public pg_data_t allows):
the 16-bit halves of k_post_part_lds' count word must be emptied on the way (k_stage_b.hpp).  Prints the md5 of the --bed=raw output of the backend named on the command line.
Run with PANGENE_POST=lds (the LDS form on a shard this small)."""
import ctypes as C
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pangene_amd import capi  # noqa: E402


class Hit(C.Structure):
    _fields_ = [("pid", C.c_int32), ("qs", C.c_int32), ("qe", C.c_int32), ("cid", C.c_int32), ("mlen", C.c_int32), ("blen", C.c_int32), ("lof", C.c_int32),
                ("rank", C.c_int32), ("score_ori", C.c_int32), ("score_adj", C.c_int32), ("score_dom", C.c_int32), ("n_exon", C.c_int32), ("off_exon", C.c_int32),
                ("pid_dom", C.c_int32), ("pid_dom0", C.c_int32), ("bits", C.c_uint32), ("cs", C.c_int64), ("cm", C.c_int64), ("ce", C.c_int64)]


class Genome(C.Structure):
    _fields_ = [("n_ctg", C.c_int32), ("m_ctg", C.c_int32), ("ctg", C.c_void_p), ("n_hit", C.c_int32), ("m_hit", C.c_int32), ("hit", C.POINTER(Hit)),
                ("n_exon", C.c_int32), ("m_exon", C.c_int32), ("exon", C.c_void_p), ("label", C.c_void_p)]


class Data(C.Structure):
    _fields_ = [("d_ctg", C.c_void_p), ("d_gene", C.c_void_p), ("d_prot", C.c_void_p), ("n_genome", C.c_int32), ("m_genome", C.c_int32), ("genome", C.POINTER(Genome)),
                ("n_gene", C.c_int32), ("m_gene", C.c_int32), ("gene", C.c_void_p), ("n_prot", C.c_int32), ("m_prot", C.c_int32), ("prot", C.c_void_p)]


def main(which, workdir):
    if which == "oracle":
        import oracle_host
        lib = oracle_host.load()
    else:
        lib = capi.load()
    C.c_int.in_dll(lib, "pg_verbose").value = 0
    n = 40000
    paf = os.path.join(workdir, "flood.paf")
    if not os.path.exists(paf):
        with open(paf, "w") as f:
            for k in range(n):
                st = 10 + k * 1000
                f.write("p1\t100\t0\t100\t+\tc1\t%d\t%d\t%d\t300\t300\t0\tms:i:%d\tcg:Z:100M\n" % (n * 1000 + 1000, st, st + 300, 400 - (k % 7)))
            for k in range(50):
                st = 500 + k * 1000
                f.write("q%d\t100\t0\t100\t+\tc1\t%d\t%d\t%d\t300\t300\t0\tms:i:300\tcg:Z:100M\n" % (k, n * 1000 + 1000, st, st + 300))
    out = os.path.join(workdir, "flood_%s.bed" % which)
    opt = capi.parse_args(lib, ["--bed=raw"])
    lib.pg_set_output(out.encode())
    d = lib.pg_data_init()
    try:
        capi.read_files(lib, opt, d, [paf])
        g = C.cast(d, C.POINTER(Data)).contents.genome[0]
        assert g.n_hit == n + 50
        for i in range(g.n_hit):
            g.hit[i].rank = 0
        lib.pg_post_process(C.byref(opt), d)
        assert lib.pg_last_error() == 0
        lib.pg_write_bed(d, 0)
    finally:
        lib.pg_data_destroy(d)
        lib.pg_set_output(None)
    print("md5", hashlib.md5(open(out, "rb").read()).hexdigest())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
