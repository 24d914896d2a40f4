"""The host half of the BVH builder (moonshine_amd/csrc/bvh_topdown.h: the top-down surface-area build over PLOC's clusters and the collapse-cost
tables) compiled with g++ and checked without a GPU: every element in the tree exactly once, every node id written exactly once, boxes that contain their
children, collapse tables that follow the recurrence, a bounded recursion on inputs a surface-area sweep cannot split, and the same tree on every run."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEAF = 0x80000000


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("topdown") / "libtopdown_shim.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall", "-o", out, os.path.join(ROOT, "tests", "shim", "topdown_shim.cpp")], check=True)
    L = C.CDLL(out)
    L.topdown_build.restype = C.c_int
    return L


def build(L, boxes, counts=None, id_base=0):
    boxes = np.ascontiguousarray(boxes, np.float32).reshape(-1, 6)
    n = len(boxes)
    counts = np.ones(n, np.uint32) if counts is None else np.ascontiguousarray(counts, np.uint32)
    left = np.zeros(n - 1, np.uint32); right = np.zeros(n - 1, np.uint32); nb = np.zeros((n - 1, 6), np.float32)
    cost = np.zeros((n - 1, 7), np.float32); split = np.zeros((n - 1, 8), np.uint8)
    root = C.c_uint32(); deepest = C.c_uint32()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = L.topdown_build(p(boxes), p(counts), n, id_base, p(left), p(right), p(nb), p(cost), p(split), C.byref(root), C.byref(deepest))
    assert rc == 0
    return dict(left=left, right=right, box=nb, cost=cost, split=split, root=root.value, deepest=deepest.value, n=n, id_base=id_base, leaf_box=boxes)


def area(b):
    d = b[..., 3:] - b[..., :3]
    return d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0]


def check_tree(t):
    n, base = t["n"], t["id_base"]
    seen_leaf = np.zeros(n, np.int64); seen_node = np.zeros(n - 1, np.int64)
    def box_of(ref):
        return t["leaf_box"][ref & ~LEAF] if ref & LEAF else t["box"][ref - base]
    def table_of(ref):
        return np.zeros(7, np.float32) if ref & LEAF else t["cost"][ref - base]
    stack = [t["root"]]
    while stack:
        r = stack.pop()
        if r & LEAF:
            seen_leaf[r & ~LEAF] += 1
            continue
        i = r - base
        assert 0 <= i < n - 1
        seen_node[i] += 1
        l, rr = int(t["left"][i]), int(t["right"][i])
        bl, br, b = box_of(l), box_of(rr), t["box"][i]
        assert np.array_equal(b[:3], np.minimum(bl[:3], br[:3])) and np.array_equal(b[3:], np.maximum(bl[3:], br[3:]))   # the union, exactly
        # collapse tables: cost(n, 1) = area + spread(8); cost(n, j) = min(spread(j), cost(n, j - 1)); split[j - 1] = slots of the left child or 0
        cl, cr = table_of(l), table_of(rr)
        def spread(j):
            return min(np.float32(cl[k - 1] + cr[j - k - 1]) for k in range(1, j) if k <= 7 and j - k <= 7)
        c = t["cost"][i]; sp = t["split"][i]
        assert c[0] == np.float32(np.float32(area(b)) + spread(8))
        for j in range(2, 8):
            assert c[j - 1] == min(spread(j), c[j - 2])
            assert (sp[j - 1] == 0) == (not spread(j) < c[j - 2])
        assert 1 <= sp[7] <= 7
        stack += [l, rr]
    assert (seen_leaf == 1).all() and (seen_node == 1).all()


def random_boxes(rs, n, spread=10.0, size=0.5):
    c = rs.uniform(-spread, spread, (n, 3)); h = rs.uniform(0.0, size, (n, 3))
    return np.concatenate([c - h, c + h], 1).astype(np.float32)


@pytest.mark.parametrize("n", [2, 3, 17, 1000])
def test_tree_is_complete_and_tables_follow_the_recurrence(shim, n):
    rs = np.random.default_rng(n)
    t = build(shim, random_boxes(rs, n), rs.integers(1, 300, n), id_base=5 * n)
    check_tree(t)
    assert t["deepest"] <= 4 * int(np.ceil(np.log2(n))) + 2


def test_ids_are_used_in_the_order_given_children_before_parents(shim):
    t = build(shim, random_boxes(np.random.default_rng(3), 500), id_base=1000)
    for i in range(t["n"] - 1):
        for ref in (int(t["left"][i]), int(t["right"][i])):
            assert ref & LEAF or ref - 1000 < i           # a PLOC-style numbering: what k_collapse and the cost tables rely on
    assert t["root"] == 1000 + t["n"] - 2


def test_identical_boxes_split_evenly(shim):
    b = np.tile(np.array([[0, 0, 0, 1, 1, 0]], np.float32), (4096, 1))
    t = build(shim, b)
    check_tree(t)
    assert t["deepest"] == 12                               # 4096 equal-cost splits: halves, not one element per level


def test_lopsided_input_stays_shallow(shim):
    """sizes in geometric progression along a line: the cheapest split peels a few large boxes off the rest level after level; the depth guard halves from level 64 on"""
    k = np.arange(2000)
    size = 0.985 ** k; pos = np.cumsum(np.concatenate([[0.0], size[:-1] * 1.1]))
    b = np.stack([pos, np.zeros_like(pos), np.zeros_like(pos), pos + size, size, size], 1).astype(np.float32)
    t = build(shim, b)
    check_tree(t)
    assert t["deepest"] <= 64 + 11


def test_surface_area_sweep_separates_clusters(shim):
    """two groups of boxes far apart end up under different children of the root, whatever their order in the input"""
    rs = np.random.default_rng(8)
    a = random_boxes(rs, 300, spread=1.0); b = random_boxes(rs, 200, spread=1.0); b[:, [0, 3]] += 100.0
    boxes = np.concatenate([a, b]); perm = rs.permutation(500); boxes = boxes[perm]
    t = build(shim, boxes)
    check_tree(t)
    def leaves(ref):
        out, st = [], [ref]
        while st:
            r = st.pop()
            if r & LEAF: out.append(r & ~LEAF)
            else: st += [int(t["left"][r]), int(t["right"][r])]
        return out
    root = t["root"]
    for side in (int(t["left"][root]), int(t["right"][root])):
        far = boxes[leaves(side), 0] > 50.0
        assert far.all() or not far.any()


def test_same_tree_every_time(shim):
    b = random_boxes(np.random.default_rng(5), 3000)
    b[100:400] = b[100]                                     # with a pile of ties in it
    t1, t2 = build(shim, b), build(shim, b)
    for k in ("left", "right", "box", "cost", "split"):
        assert np.array_equal(t1[k], t2[k])
