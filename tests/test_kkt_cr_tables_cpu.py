"""The tables of the chain's cyclic reduction (csrc/pc_kkt_cr.hpp through ``pc_kkt_cr_plan``; host integer work, no GPU)
against a symbolic elimination of the chain graph: eliminating the nodes in the tables' level order, a node's neighbours at
the moment it goes must be exactly its separators, the fill edge between them must have been created by the node the
tables name as the coupling's source, and every eliminated node must stand in the pull lists of exactly its separators --
for plain segments and for segments whose first and / or last node is exported (pycollo_amd/kkt_sharded.py)."""
import ctypes as C

import numpy as np
import pytest


def cr_plan(lib, seg_lengths, export):
    n_chain = int(sum(seg_lengths))
    seg_ptr = np.concatenate([[0], np.cumsum(seg_lengths)]).astype(np.int64)
    chain_ptr = (np.arange(n_chain + 1) * 3).astype(np.int64)
    exp = np.ascontiguousarray(export, dtype=np.uint8)
    a, b, ma, mb = (np.zeros(max(1, n_chain), np.int64) for _ in range(4))
    lvl = np.zeros(max(1, n_chain), np.int32)
    pp = np.zeros(n_chain + 1, np.int64)
    pe = np.zeros(max(1, 2 * n_chain), np.int32)
    lib.pc_kkt_cr_plan.argtypes = [C.c_int64, C.c_int64] + [C.c_void_p] * 2 + [C.c_int64] + [C.c_void_p] * 8 + [C.c_int64]
    lib.pc_kkt_last_error.restype = C.c_char_p
    ok = lib.pc_kkt_cr_plan(n_chain, len(seg_lengths), seg_ptr.ctypes.data, chain_ptr.ctypes.data, 2, exp.ctypes.data if exp.any() else None,
                            a.ctypes.data, b.ctypes.data, ma.ctypes.data, mb.ctypes.data, lvl.ctypes.data, pp.ctypes.data, pe.ctypes.data, len(pe))
    if not ok:
        raise RuntimeError(lib.pc_kkt_last_error().decode())
    return seg_ptr, a[:n_chain], b[:n_chain], ma[:n_chain], mb[:n_chain], lvl[:n_chain], pp, pe


def check(seg_ptr, export, a, b, ma, mb, lvl, pp, pe):
    n_chain = int(seg_ptr[-1])
    pulls = {c: [(int(x) >> 1, int(x) & 1) for x in pe[pp[c]:pp[c + 1]]] for c in range(n_chain)}
    expect_pulls = {c: [] for c in range(n_chain)}
    for s0, s1 in zip(seg_ptr[:-1], seg_ptr[1:]):
        nodes = list(range(int(s0), int(s1)))
        nbr = {c: set() for c in nodes}
        for c in nodes[:-1]:
            nbr[c].add(c + 1); nbr[c + 1].add(c)
        made = {}                                      # fill edge -> the node whose elimination created it last
        for c in sorted(nodes, key=lambda c: (lvl[c], c)):
            live = sorted(nbr[c])
            want = sorted(x for x in (int(a[c]), int(b[c])) if x >= 0)
            if export[c]:
                # what is left when an exported node's turn comes is the segment's other exported node, if there is one; the
                # coupling of the two is kept once, in the first node's panel
                others = [x for x in nodes if export[x] and x != c]
                assert live == others, (c, live, others)
                assert want == ([x for x in others if x > c]), (c, want)
            else:
                assert live == want, (c, live, want)   # the separators are the node's neighbours when its turn comes
            for sep, mid in ((int(a[c]), int(ma[c])), (int(b[c]), int(mb[c]))):
                if sep >= 0:
                    key = (min(sep, c), max(sep, c))
                    assert mid == made.get(key, -1), (c, sep, mid, made.get(key))
                    assert mid >= 0 or abs(sep - c) == 1
            if export[c]:
                continue                               # stays: nothing is eliminated into its neighbours
            assert int(a[c]) < c and (int(b[c]) < 0 or int(b[c]) > c)
            for sep, side in ((int(b[c]), 0), (int(a[c]), 1)):
                if sep >= 0:
                    expect_pulls[sep].append((c, side))
            for x in live:
                nbr[x].discard(c)
            if len(live) == 2:
                nbr[live[0]].add(live[1]); nbr[live[1]].add(live[0])
                made[(live[0], live[1])] = c
            nbr[c] = set()
        # what is left of the segment: its exported nodes, coupled to each other if there are two
        left = [c for c in nodes if export[c]]
        assert all(nbr[c] <= set(left) for c in left)
    for c in range(n_chain):
        assert sorted(pulls[c]) == sorted(expect_pulls[c]), c
        lv = [lvl[e] for e, _ in pulls[c]]
        assert lv == sorted(lv)                        # level by level: the order the terms are added in
        assert len(pulls[c]) <= 64


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 8, 9, 16, 17, 31, 33, 100, 257, 1000])
def test_tables_describe_the_elimination_of_the_chain_graph(built, n):
    from pycollo_amd.engine import load_library
    lib = load_library()
    for first, last in ((0, 0), (1, 0), (0, 1), (1, 1)):
        if n < 2 and (first or last):
            continue
        export = np.zeros(2 * n + 3, np.uint8)         # two segments of n nodes and one of 3 in between: ids must not mix
        export[0], export[n - 1] = first, last
        export[n + 3], export[2 * n + 2] = last, first
        seg_ptr, *tabs = cr_plan(lib, [n, 3, n], export)
        check(seg_ptr, export, *tabs)


def test_an_exported_node_inside_a_segment_is_refused(built):
    from pycollo_amd.engine import load_library
    export = np.zeros(5, np.uint8)
    export[2] = 1
    with pytest.raises(RuntimeError, match="first and the last"):
        cr_plan(load_library(), [5], export)


def test_the_table_builder_under_asan_and_ubsan(built, tmp_path):
    """``pc_kkt_cr.hpp`` compiled on its own with ``-fsanitize=address,undefined`` (tests/c/kkt_cr_sanitize.cpp): no
    sanitizer report, and the same tables as the library's."""
    import os
    import subprocess
    from conftest import ROOT
    from pycollo_amd.engine import load_library
    exe = os.path.join(ROOT, "tests", "_build", "kkt_cr_sanitize")
    src = os.path.join(ROOT, "tests", "c", "kkt_cr_sanitize.cpp")
    hdr = os.path.join(ROOT, "pycollo_amd", "csrc", "pc_kkt_cr.hpp")
    if not os.path.exists(exe) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(exe):
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        res = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                              "-fno-omit-frame-pointer", "-o", exe + f".tmp{os.getpid()}", src], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[-3000:]
        os.replace(exe + f".tmp{os.getpid()}", exe)
    lib = load_library()
    for segs in ([(1, 0, 0)], [(2, 1, 1)], [(7, 0, 1), (3, 1, 1), (12, 1, 0)], [(1000, 1, 1), (513, 0, 0)], [(5, 0, 0), (1, 0, 0), (2, 0, 1)]):
        args = [str(x) for s in segs for x in s]
        res = subprocess.run([exe, "2", "3"] + args, capture_output=True, text=True, timeout=120)
        assert res.returncode == 0 and not res.stderr.strip(), res.stderr[-2000:]
        out = {ln.split()[0]: [int(v) for v in ln.split()[1:]] for ln in res.stdout.splitlines() if ln and not ln.startswith("buf_len")}
        export = np.concatenate([[int(p == 0 and f) or int(p == n - 1 and l) for p in range(n)] for n, f, l in segs]).astype(np.uint8)
        seg_ptr, a, b, ma, mb, lvl, pp, pe = cr_plan(lib, [n for n, _, _ in segs], export)
        for name, arr in (("a", a), ("b", b), ("mid_a", ma), ("mid_b", mb), ("level", lvl), ("pull_ptr", pp), ("pull_e", pe[:pp[-1]])):
            assert out.get(name, []) == [int(v) for v in arr], name
