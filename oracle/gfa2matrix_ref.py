#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- plain-Python restatement of `pangene.js gfa2matrix` (reference pangene.js:1168-1247, with the
GFA parser it relies on, pangene.js:131-197), the checker for pg_write_matrix / pg_gfa2matrix_file.

Pinning: the reference script needs the k8 JavaScript runtime, which is absent in this image, so this restatement could not be
run against it ("parity unpinned" for this row); the tests anchor it on hand-countable entries of the reference's own test/C4
data instead.  k8's print() joins its arguments with a TAB.

    python oracle/gfa2matrix_ref.py [-c] [-d clstr] in.gfa
"""
import gzip
import re
import sys


def _lines(fn):
    op = gzip.open if fn.endswith(".gz") else open
    with op(fn, "rt") as f:
        return f.read().split("\n")


def gfa2matrix(gfa_lines, copy_number=False, clstr_lines=None, print_cd=False):
    seg, segname = [], {}          # pangene.js:131-135 (#seg_add): ids in the order S- and L-lines introduce the names

    def seg_add(name):
        if name not in segname:
            segname[name] = len(seg)
            seg.append(name)
        return segname[name]

    walks = []
    for line in gfa_lines:
        if not line:
            continue
        t = line.split("\t")
        if line[0] == "S":             # pangene.js:131-145
            if len(t) >= 3:
                seg_add(t[1])
        elif line[0] == "L":           # pangene.js:146-168
            if len(t) >= 5 and t[2] in "+-" and t[4] in "+-" and len(t[2]) == 1 and len(t[4]) == 1:
                seg_add(t[1]); seg_add(t[3])
        elif line[0] == "W":           # pangene.js:169-192: steps whose name is not a segment (yet) are dropped
            if len(t) < 7:
                continue
            v = [segname[m.group(2)] for m in re.finditer(r"([><])([^\s><]+)", t[6]) if m.group(2) in segname]
            walks.append((t[1] + "#" + t[2], v))
    asm_h, asm_a = {}, []              # pangene.js:1184-1190
    for a, _ in walks:
        if a not in asm_h:
            asm_h[a] = len(asm_a)
            asm_a.append(a)
    mat = [[0] * len(asm_a) for _ in seg]
    for a, v in walks:                 # pangene.js:1193-1197
        for s in v:
            mat[s][asm_h[a]] += 1
    out, paralog = [], {}
    if clstr_lines is not None:        # pangene.js:1198-1234
        b = []

        def process(b):
            sel = -1
            for i, (_, star) in enumerate(b):
                if star:
                    sel = i
            if sel >= 0:
                for i, (name, _) in enumerate(b):
                    if i != sel:
                        paralog[name.split(":")[0]] = b[sel][0].split(":")[0]
                        if print_cd:
                            out.append(name.split(":")[0] + "\t" + b[sel][0].split(":")[0])
        for line in clstr_lines:
            if line.startswith(">"):
                process(b); b = []
            else:
                m = re.match(r"^\d+\s+\S+,\s+>(\S+)\.\.\.\s+(\S+)", line)
                if m:
                    b.append((m.group(1), m.group(2) == "*"))
        process(b)
        def js_key_order(keys):  # `for (const g in paralog)`: integer-like keys first, ascending; then the others in insertion order
            ints = sorted((int(k), k) for k in keys if re.fullmatch(r"0|[1-9][0-9]{0,9}", k) and int(k) < 4294967295)
            isint = {k for _, k in ints}
            return [k for _, k in ints] + [k for k in keys if k not in isint]
        for g in js_key_order(list(paralog)):
            p = paralog[g]
            if g in segname and p in segname:
                for i in range(len(asm_a)):
                    mat[segname[p]][i] += mat[segname[g]][i]
    if print_cd:
        return "".join(x + "\n" for x in out)
    if not copy_number:                # pangene.js:1235-1239
        mat = [[1 if x > 1 else x for x in r] for r in mat]
    out.append("Gene\t" + "\t".join(asm_a))
    for i, r in enumerate(mat):        # pangene.js:1242-1245
        if seg[i] not in paralog:
            out.append(seg[i] + "\t" + "\t".join(str(x) for x in r))
    return "".join(x + "\n" for x in out)


if __name__ == "__main__":
    args = sys.argv[1:]
    cn = "-c" in args
    if cn:
        args.remove("-c")
    cl = None
    if "-d" in args:
        i = args.index("-d")
        cl = _lines(args[i + 1]); del args[i:i + 2]
    sys.stdout.write(gfa2matrix(_lines(args[0]), cn, cl))
