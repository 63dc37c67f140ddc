"""Host logic of the N x M compare block (no GPU): given the connected component of every row and
column, the plan must (a) order rows / columns so that components are contiguous and stable,
(b) list every tile that holds a same-component pair -- or, in symmetric mode, the tile holding
its mirror -- and (c) list nothing twice.  A tile missing from the list would silently turn
sharing pairs into zeros, hence this test."""
import ctypes as C

import numpy as np
import pytest


def _plan(lib, comp_r, comp_c, tr, tc, symmetric):
    comp_r = np.ascontiguousarray(comp_r, dtype=np.uint32)
    comp_c = np.ascontiguousarray(comp_c, dtype=np.uint32)
    nr, nc = comp_r.size, comp_c.size
    mx = int(max(comp_r.max(initial=0), comp_c.max(initial=0))) + 1
    rperm, cperm = np.zeros(max(nr, 1), np.uint32), np.zeros(max(nc, 1), np.uint32)
    cap = ((nr + tr - 1) // tr) * ((nc + tc - 1) // tc) + 1
    tiles = np.zeros(2 * cap, np.uint32)
    n = C.c_uint32()
    p32 = C.POINTER(C.c_uint32)
    rc = lib.smh_test_plan_tiles(comp_r.ctypes.data_as(p32), nr, comp_c.ctypes.data_as(p32), nc, mx, tr, tc, symmetric,
                                 rperm.ctypes.data_as(p32), cperm.ctypes.data_as(p32), tiles.ctypes.data_as(p32), cap,
                                 C.byref(n))
    assert rc == 0
    return rperm[:nr], cperm[:nc], tiles[: 2 * n.value].reshape(-1, 2)


@pytest.mark.parametrize("seed", range(6))
def test_plan_covers_every_same_component_pair(pkg_lib, seed):
    rng = np.random.RandomState(seed)
    nr, nc = rng.randint(1, 300), rng.randint(1, 400)
    ncomp = rng.choice([1, 3, 40, 1000])
    comp_r = rng.randint(0, ncomp, nr) * 2            # row nodes: even ids ...
    comp_c = rng.randint(0, ncomp, nc) * 2
    comp_c[rng.rand(nc) < 0.2] += 1                   # ... some columns in components no row has
    tr, tc = int(rng.choice([4, 8, 16])), 64
    rperm, cperm, tiles = _plan(pkg_lib, comp_r, comp_c, tr, tc, False)
    assert sorted(rperm) == list(range(nr)) and sorted(cperm) == list(range(nc))
    # contiguous by component, original order inside a component
    for perm, comp in ((rperm, comp_r), (cperm, comp_c)):
        keys = [(comp[i], i) for i in perm]
        assert keys == sorted(keys)
    listed = {(int(a), int(b)) for a, b in tiles}
    assert len(listed) == len(tiles)
    rslot, cslot = np.argsort(rperm), np.argsort(cperm)
    need = {(int(rslot[i]) // tr, int(cslot[j]) // tc) for i in range(nr) for j in range(nc) if comp_r[i] == comp_c[j]}
    assert need <= listed
    # nothing beyond the bounding boxes (row slot range x column slot range) of the shared components
    box = set()
    for c in set(comp_r.tolist()) & set(comp_c.tolist()):
        rs = rslot[comp_r == c]; cs = cslot[comp_c == c]
        for ti in range(int(rs.min()) // tr, int(rs.max()) // tr + 1):
            for tj in range(int(cs.min()) // tc, int(cs.max()) // tc + 1):
                box.add((ti, tj))
    assert listed == box


@pytest.mark.parametrize("seed", range(6))
def test_symmetric_plan_covers_each_pair_or_its_mirror(pkg_lib, seed):
    rng = np.random.RandomState(100 + seed)
    n = rng.randint(1, 500)
    comp = rng.randint(0, rng.choice([1, 5, 60, 2000]), n)
    tr, tc = int(rng.choice([4, 8, 16])), 64
    rperm, cperm, tiles = _plan(pkg_lib, comp, comp, tr, tc, True)
    assert (rperm == cperm).all()
    listed = {(int(a), int(b)) for a, b in tiles}
    slot = np.argsort(rperm)
    for i in range(n):
        same = np.nonzero(comp == comp[i])[0]
        for j in same:
            a, b = int(slot[i]), int(slot[j])
            assert (a // tr, b // tc) in listed or (b // tr, a // tc) in listed, (i, j)
    # no tile wholly below the diagonal is launched
    for ti, tj in listed:
        assert not (tj * tc + tc - 1 < ti * tr)
