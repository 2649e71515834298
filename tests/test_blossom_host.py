"""The product's blossom restatement (squarna_amd/csrc/sq_blossom.h), compiled for the host, against
networkx.max_weight_matching on random graphs with many ties (the reference's a-9 dependency)."""
import os
import random
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "blossom_host.cpp")
EXE = os.path.join(HERE, "native", "_build", "blossom_host")


@pytest.fixture(scope="module")
def exe():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", EXE, SRC])
    return EXE


def _graphs(seed, count, nmax=22, mmax=60):
    rng = random.Random(seed)
    out = []
    for _ in range(count):
        n = rng.randint(2, nmax)
        m = rng.randint(1, min(mmax, n * (n - 1) // 2))
        pairs = set()
        while len(pairs) < m:
            a, b = rng.sample(range(n), 2)
            pairs.add((min(a, b), max(a, b)))
        pairs = list(pairs)
        rng.shuffle(pairs)
        small = rng.random() < 0.6                        # few distinct weights -> non-unique optima
        edges = [(a, b, float(rng.randint(1, 4)) if small else round(rng.uniform(1, 30), 3) ** 1.7) for a, b in pairs]
        out.append(edges)
    return out


@pytest.mark.parametrize("seed,count,nmax,mmax", [(7, 400, 22, 60), (11, 60, 120, 700)])
def test_blossom_matches_networkx(exe, seed, count, nmax, mmax):
    """(the second set: graphs of the size of SRtest150's -- nested blossoms, expansions, lists scanned in chunks)"""
    nx = pytest.importorskip("networkx")
    graphs = _graphs(seed, count, nmax, mmax)
    lines, idmaps = [str(len(graphs))], []
    for edges in graphs:
        ids = {}
        for a, b, _ in edges:                              # vertex ids in first-appearance order == nx node order
            ids.setdefault(a, len(ids))
            ids.setdefault(b, len(ids))
        idmaps.append(ids)
        lines.append("%d %d" % (len(ids), len(edges)))
        lines += ["%d %d %r" % (ids[a], ids[b], w) for a, b, w in edges]
    res = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True)
    rows = res.stdout.strip().split("\n")
    assert len(rows) == len(graphs)
    for edges, ids, row in zip(graphs, idmaps, rows):
        G = nx.Graph()
        for a, b, w in edges:
            G.add_edge(a, b, weight=w)
        exp = nx.max_weight_matching(G)
        nums = list(map(int, row.split()))
        n = len(nums) // 2
        mate, mord = nums[:n], nums[n:]
        inv = {v: k for k, v in ids.items()}
        # networkx returns each pair as (u, v) with u the endpoint that entered its `mate` dict first
        # (matching_dict_to_set); Edmonds() sorts those tuples (SQRNalgos.py:109), so the orientation is observable
        got = {(inv[v], inv[mate[v]]) for v in range(n) if mate[v] >= 0 and mord[v] < mord[mate[v]]}
        assert got == exp, (edges, got, exp)
