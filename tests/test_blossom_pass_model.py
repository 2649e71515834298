"""The rules by which sq_blossom.h scans SEVERAL queue vertices in one pass, as an executable model (CPU only).

networkx's max_weight_matching (3.4.2; the reference's SQRNalgos.py:96-110 calls it) scans the neighbours of one S-vertex
after the other.  The kernel lets the neighbour lists of up to four queue vertices share the 64 lanes of a pass: every lane
classifies its neighbour against the state BEFORE the pass, the lanes in front of the first state-changing neighbour (an
"event") are applied together, and what two lists can share is resolved so that the state after the pass is the state
the sequential scan leaves:
  * cat 2 (an allowed edge into a T-blossom, w unlabelled): the first lane in scan order labels w; later lanes on w --
    another cat 2, or a best-edge competitor for w -- do nothing.  Kernel: a marker 8 + segment in the label byte, stored
    in reverse segment order, read back by the cat-3 lanes, then T.
  * cat 3 / cat 4 (best edge of w / of the scanning vertex's blossom): "the first strictly smaller slack wins" == the
    minimum slack, lowest lane among ties.  Kernel: ds_min_u64 on the cached slack (+inf where there is no best edge),
    read back, the tying lanes store the edge in reverse segment order (per blossom and segment: the lowest lane only).
Stores of ONE instruction land in an order the hardware chooses: the model shuffles them, so a rule that depended on it
would show.  This checks the rules (the kernel itself is checked against networkx on the GPU: test_hip_parity3.py,
tools/fuzz_parity.py); it is the place to try a change of the rules first.
"""
import math
import random

INF = math.inf


class State:
    def __init__(self, rng):
        n = self.n = rng.randrange(6, 26)
        # top-level blossoms: some vertices alone (blossom id == vertex id), some grouped under an id >= n
        self.inblossom = list(range(n))
        nb = n
        verts = list(range(n))
        rng.shuffle(verts)
        while verts and rng.random() < 0.6:
            size = rng.choice((3, 3, 5))
            grp, verts = verts[:size], verts[size:]
            for v in grp:
                self.inblossom[v] = nb
            nb += 1
        self.nb = nb
        self.label = [0] * nb
        for b in set(self.inblossom):
            self.label[b] = rng.choice((0, 1, 1, 2, 2))
        for v in range(n):
            b = self.inblossom[v]
            if b != v:                                   # a vertex inside a blossom: S-blossoms label their vertices S,
                self.label[v] = 1 if self.label[b] == 1 else (rng.choice((0, 2)) if self.label[b] == 2 else 0)
        self.labeledge = [-1] * nb
        dvals = rng.choice(((3.0,), (2.0, 3.0), (1.5, 2.0, 2.5, 3.0)))
        self.dual = [rng.choice(dvals) for _ in range(n)]
        # edges with few distinct weights: ties everywhere
        self.edges = []
        seen = set()
        for _ in range(rng.randrange(n, 4 * n)):
            v, w = rng.sample(range(n), 2)
            if (min(v, w), max(v, w)) in seen:
                continue
            seen.add((min(v, w), max(v, w)))
            self.edges.append((v, w, rng.choice((0.5, 1.0, 1.5, 2.0, 2.5, 3.0))))
        self.adj = [[] for _ in range(n)]                # (de, w, weight) in insertion order
        for e, (v, w, wt) in enumerate(self.edges):
            self.adj[v].append((2 * e, w, wt))
            self.adj[w].append((2 * e + 1, v, wt))
        self.allow = [False] * len(self.edges)
        for e, (v, w, wt) in enumerate(self.edges):
            if self.dual[v] + self.dual[w] - 2 * wt <= 0 and rng.random() < 0.5:
                self.allow[e] = True                     # (tight edges, some already allowed)
        self.bestedge = [-1] * nb
        self.bslack = [INF] * nb
        for x in range(nb):
            if rng.random() < 0.4 and self.edges:
                e = rng.randrange(len(self.edges))
                v, w, wt = self.edges[e]
                s = self.dual[v] + self.dual[w] - 2 * wt
                if s > 0:
                    self.bestedge[x], self.bslack[x] = 2 * e, s

    def key(self):
        return (tuple(self.label), tuple(self.labeledge), tuple(self.bestedge), tuple(self.bslack), tuple(self.allow))

    def clone(self):
        import copy
        return copy.deepcopy(self)


def lanes_of(st, segs):
    out = []
    for s, v in enumerate(segs):
        for de, w, wt in st.adj[v]:
            out.append((s, v, de, w, wt))
    return out


def sequential(st, lanes):
    """networkx's neighbour loop over the lanes in order; returns the index of the first event (len(lanes): none)."""
    for k, (s, v, de, w, wt) in enumerate(lanes):
        bv, bw = st.inblossom[v], st.inblossom[w]
        if bv == bw:
            continue
        e = de >> 1
        ks = None
        allowed = st.allow[e]
        if not allowed:
            ks = st.dual[v] + st.dual[w] - 2 * wt
            allowed = ks <= 0
        if allowed:
            if st.label[bw] in (0, 1):
                return k                                 # assignLabel / scanBlossom: state-changing (the event's own work,
            st.allow[e] = True                           # its allowedge included, is not the pass's)
            if st.label[w] == 0:
                st.label[w] = 2
                st.labeledge[w] = de
        elif st.label[bw] == 1:
            if st.bestedge[bv] == -1 or ks < st.bslack[bv]:
                st.bestedge[bv], st.bslack[bv] = de, ks
        elif st.label[w] == 0:
            if st.bestedge[w] == -1 or ks < st.bslack[w]:
                st.bestedge[w], st.bslack[w] = de, ks
    return len(lanes)


def one_pass(st, lanes, nseg, rng):
    """The kernel's pass: classification against the state before the pass, then the ordered application."""
    cls = []
    for lane, (s, v, de, w, wt) in enumerate(lanes):
        bv, bw = st.inblossom[v], st.inblossom[w]
        was = st.allow[de >> 1]
        ks = st.dual[v] + st.dual[w] - 2 * wt
        cons = bw != bv
        becomes = cons and not was and ks <= 0
        allowed = was or becomes
        lbw, lw = st.label[bw], st.label[w]
        cat = 0
        if cons:
            if allowed:
                cat = 1 if lbw in (0, 1) else (2 if lw == 0 else 0)
            else:
                cat = 4 if lbw == 1 else (3 if lw == 0 else 0)
        cls.append(dict(lane=lane, seg=s, v=v, de=de, w=w, bv=bv, bw=bw, ks=ks, becomes=becomes, cat=cat, lbw=lbw,
                        s_bew=st.bslack[w], s_bebv=st.bslack[bv]))
    f = next((c["lane"] for c in cls if c["cat"] == 1), len(lanes))
    act = [c for c in cls if c["lane"] < f]

    def instruction(stores):                            # one store instruction: its lanes land in hardware order
        stores = list(stores)
        rng.shuffle(stores)
        for fn in stores:
            fn()
    instruction([(lambda c=c: st.allow.__setitem__(c["de"] >> 1, True)) for c in act if c["becomes"]])
    c2 = [c for c in act if c["cat"] == 2]
    c3 = [c for c in act if c["cat"] == 3 and c["ks"] < c["s_bew"]]
    c4 = [c for c in act if c["cat"] == 4 and c["ks"] < c["s_bebv"]]
    if c2:
        for s in range(nseg - 1, -1, -1):
            instruction([(lambda c=c: (st.label.__setitem__(c["w"], 8 + c["seg"]), st.labeledge.__setitem__(c["w"], c["de"])))
                         for c in c2 if c["seg"] == s])
        if any(c["lbw"] == 2 for c in c3):
            c3 = [c for c in c3 if not (st.label[c["w"]] >= 8 and st.label[c["w"]] - 8 < c["seg"])]
        instruction([(lambda c=c: st.label.__setitem__(c["w"], 2)) for c in c2])
    comp = [(c, c["w"]) for c in c3] + [(c, c["bv"]) for c in c4]
    if comp:
        for c, x in comp:                                # ds_min_u64: order-free
            st.bslack[x] = min(st.bslack[x], c["ks"])
        tie = [(c, x) for c, x in comp if st.bslack[x] == c["ks"]]
        for s in range(nseg - 1, -1, -1):
            ts = [(c, x) for c, x in tie if c["seg"] == s]
            t4 = [c["lane"] for c, x in ts if c["cat"] == 4]
            l4 = min(t4) if t4 else -1
            instruction([(lambda c=c, x=x: st.bestedge.__setitem__(x, c["de"])) for c, x in ts if c["cat"] == 3 or c["lane"] == l4])
    return f


def test_several_vertices_per_pass_leave_the_sequential_state():
    rng = random.Random(99)
    stats = dict(passes=0, events=0, shared_w=0, c2=0, ties=0)
    for trial in range(3000):
        st = State(rng)
        svert = [v for v in range(st.n) if st.label[st.inblossom[v]] == 1 and st.adj[v]]
        if not svert:
            continue
        nseg = min(len(svert), rng.randrange(1, 5))
        segs = rng.sample(svert, nseg)
        if rng.random() < 0.05 and nseg > 1:
            segs[-1] = segs[0]                           # the same vertex twice in the queue
        lanes = lanes_of(st, segs)
        if not lanes or len(lanes) > 64:
            continue
        a, b = st.clone(), st.clone()
        fa = sequential(a, lanes)
        fb = one_pass(b, lanes, nseg, rng)
        assert fa == fb, (trial, fa, fb)
        assert a.key() == b.key(), (trial, segs, [(x, y) for x, y in zip(a.key(), b.key()) if x != y])
        stats["passes"] += 1
        stats["events"] += fa < len(lanes)
        ws = [w for s, v, de, w, wt in lanes[:fa]]
        stats["shared_w"] += len(ws) != len(set(ws))
        stats["c2"] += any(x == 2 and y != 2 for x, y in zip(a.label, st.label))
        stats["ties"] += any(x != y for x, y in zip(a.bestedge, st.bestedge))
    # the generator really reaches the cases the rules are for
    assert stats["passes"] > 2000 and stats["shared_w"] > 300 and stats["c2"] > 100 and stats["ties"] > 500, stats
