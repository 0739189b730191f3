"""CPU-only static checks of the RECORDED multi-GPU programs (gpexp_amd.dist.Program): the 2-D panel loop is run against the
recorder for EVERY rank of a grid -- no device, fake handles -- and the recorded rows are analysed:

  * event discipline per rank: every WAIT refers to an event that an EARLIER row of the same program recorded (a wait on a
    never-recorded event would order nothing in the first step), never one recorded on the waiting stream itself; every side
    stream starts behind the fork event and is joined into MAIN at the end (what a stream capture needs);
  * collective order across ranks: for every communicator (world, each process row, each process column) all member ranks
    issue the same sequence of collectives with the same counts and roots -- the property that makes the loop deadlock-free
    under RCCL, which has never run with more than one rank on the build's hardware;
  * buffer ring: a packed panel buffer is rewritten only after waits on everything that read it a ring earlier.
"""
import ctypes

import pytest

from gpexp_amd import dist
from gpexp_amd.dist import OP

NAMES = {v: k for k, v in OP.items()}


class FakeMat:
    _n = [1000]

    def __init__(self):
        FakeMat._n[0] += 8
        self.h = ctypes.c_void_p(FakeMat._n[0])


def record_rank(n, nb, Pr, Pc, rank, agg, bulk, streamed, monkeypatch, window=0):
    monkeypatch.setenv("GPX_DIST_BULK", bulk)
    geo = dist.Grid2D(n, nb, Pr, Pc, rank)
    ops = dist.DeviceOps2D(None)
    comm = object.__new__(dist.RcclComm)
    comm.ctx, comm.rank, comm.world = None, rank, Pr * Pc
    prog = dist.Program()
    ops.prog = comm.prog = prog
    A, L, B = FakeMat(), FakeMat(), FakeMat()
    G = [FakeMat() for _ in range(dist.ring_size(agg))]
    hook = dist.streamed_ivar_hook(ops, geo, L, B, agg, window, stream=dist.EVAL, fwd=FakeMat() if window else None) if streamed else None
    E = [FakeMat() for _ in G]          # the product path: the early block row travels into buffers of its own
    dist.dist2_potrf_enqueue(ops, comm, geo, A, G, L=L, on_stored=hook, agg=agg, window=window, E=E)
    return geo, prog, G


def analyse_events(prog):
    cur, recorded, first_on_stream = dist.MAIN, {}, {}
    for i, r in enumerate(prog.rows):
        op = NAMES[r[0]]
        if op == "STREAM":
            cur = r[4]
            continue
        if op == "RECORD":
            recorded[r[4]] = (cur, i)
        elif op == "WAIT":
            assert r[4] in recorded, "row %d waits for event %d that no earlier row records" % (i, r[4])
            assert recorded[r[4]][0] != cur, "row %d: stream %d waits for its own event %d" % (i, cur, r[4])
        if cur not in first_on_stream:
            first_on_stream[cur] = (op, r[4])
    for s in dist.ALL_SIDE_STREAMS:
        assert first_on_stream.get(s) == ("WAIT", dist.EV_FORK), "stream %d does not start behind the fork" % s
    tail = [(NAMES[r[0]], r[4]) for r in prog.rows[-len(dist.ALL_SIDE_STREAMS):]]
    assert tail == [("WAIT", dist.EV_JOIN0 + i) for i in range(len(dist.ALL_SIDE_STREAMS))], "no join at the end"


def collectives(prog):
    """[(group, kind, count(s), root(s))] in issue order."""
    out = []
    for r in prog.rows:
        op = NAMES[r[0]]
        a = r[4:]
        if op == "BCAST_GRP":
            out.append((a[3], "bcast", a[1], a[2]))
        elif op == "BCAST_GRP2":
            out.append((a[4], "bcast", a[2], a[3]))
        elif op == "REDUCE_GRP":
            out.append((a[3], "reduce", a[1], a[2]))
        elif op == "ALLREDUCE":
            out.append((dist.WORLD, "allreduce", a[1], -1))
        elif op == "PANEL_BCAST":
            npieces, off = a[0], a[1]
            ex = prog.extra[off:off + 3 * npieces]
            out.append((dist.WORLD, "panel_bcast", tuple(ex[npieces:2 * npieces]), tuple(ex[2 * npieces:])))
    return out


@pytest.mark.parametrize("Pr,Pc,n,nb,agg,bulk", [(1, 2, 2500, 128, 4, "eval"), (2, 2, 2500, 128, 4, "eval"), (2, 4, 4200, 128, 4, "eval"),
                                                 (2, 4, 4200, 128, 8, "eval"), (2, 4, 3000, 256, 2, "chunks"), (2, 3, 2500, 128, 3, "bulk"),
                                                 (4, 2, 2500, 128, 4, "main"), (1, 1, 2500, 128, 4, "eval"), (2, 4, 32768, 512, 4, "eval")])
def test_recorded_programs_event_discipline_and_collective_order(Pr, Pc, n, nb, agg, bulk, monkeypatch):
    W = Pr * Pc
    progs = {}
    for rank in range(W):
        # from 4 ranks: the product default -- evaluation streamed against a WINDOW of the factor (two groups of block columns)
        geo, prog, G = record_rank(n, nb, Pr, Pc, rank, agg, bulk, streamed=W >= 4, monkeypatch=monkeypatch,
                                   window=2 * agg if W >= 4 else 0)
        analyse_events(prog)
        progs[rank] = collectives(prog)
    # world communicator: identical sequences on every rank
    world = {r: [c for c in seq if c[0] == dist.WORLD] for r, seq in progs.items()}
    assert all(world[r] == world[0] for r in range(W)), "ranks disagree on the order of the world collectives"
    # round 5: a panel travels in one or two row chunks (the same cut on every rank: the sequences above agree), and the chunks of
    # a panel deliver exactly the regions of its pieces
    nblk = dist.num_blocks(n, nb)
    assert nblk <= len(world[0]) <= 2 * nblk, "one or two panel broadcasts per panel"
    if nblk >= 12 and min(Pr, Pc) >= 1 and dist.Grid2D(n, nb, Pr, Pc, 0).piece_rows(0, 0) >= 4 * nb:
        assert len(world[0]) > nblk, "no panel was cut into row chunks"
    assert sum(sum(c[2]) for c in world[0]) == sum(cnt for k in range(nblk) for _, cnt, _ in geo.pieces(k)), \
        "the chunks of the panels do not add up to their pieces"
    # row / column communicators: identical sequences inside each group
    for grp, members in [(dist.ROW, [[p * Pc + q for q in range(Pc)] for p in range(Pr)]),
                         (dist.COL, [[p * Pc + q for p in range(Pr)] for q in range(Pc)])]:
        for ranks in members:
            seqs = [[c for c in progs[r] if c[0] == grp] for r in ranks]
            assert all(s == seqs[0] for s in seqs), "ranks %s disagree on the order of their group-%d collectives" % (ranks, grp)
    assert_global_collective_order(progs, Pr, Pc)


def assert_global_collective_order(progs, Pr, Pc):
    """Deadlock freedom across DIFFERENT communicators: every rank issues all its collectives on one stream, so a collective
    blocks the ones behind it.  Node = the i-th collective of a communicator (world / one process row / one process column),
    edge = "issued before" on some rank; the union over all ranks must be acyclic (a total order all ranks agree with exists).
    Per-communicator agreement alone does not give that: X: [row-op, col-op], Z: [col-op', ...] can still form a cycle."""
    def members(grp, rank):
        pr, pc = rank // Pc, rank % Pc
        if grp == dist.WORLD:
            return ("w",)
        return ("r", pr) if grp == dist.ROW else ("c", pc)
    size = {dist.WORLD: Pr * Pc, dist.ROW: Pc, dist.COL: Pr}
    succ, indeg = {}, {}
    for rank, seq in progs.items():
        count, prev = {}, None
        for c in seq:
            if size[c[0]] == 1:
                continue                                  # a one-member group issues nothing
            key = members(c[0], rank)
            i = count.get(key, 0)
            count[key] = i + 1
            node = (key, i)
            indeg.setdefault(node, 0)
            succ.setdefault(node, set())
            if prev is not None and node not in succ[prev]:
                succ[prev].add(node)
                indeg[node] += 1
            prev = node
    ready = [n for n, d in indeg.items() if d == 0]
    done = 0
    while ready:
        n = ready.pop()
        done += 1
        for m in succ[n]:
            indeg[m] -= 1
            if indeg[m] == 0:
                ready.append(m)
    assert done == len(indeg), "the ranks' collective orders form a cycle across communicators (%d of %d ordered)" % (done, len(indeg))


def test_global_collective_order_detects_a_cycle():
    """The checker itself: two ranks of a 2x2 grid that issue a row and a column collective in opposite order -- with a third
    rank closing the loop -- must be flagged."""
    ok = {0: [(dist.ROW, "bcast", 1, 0), (dist.COL, "bcast", 1, 0)], 1: [(dist.ROW, "bcast", 1, 0), (dist.COL, "bcast", 1, 0)],
          2: [(dist.ROW, "bcast", 1, 0), (dist.COL, "bcast", 1, 0)], 3: [(dist.ROW, "bcast", 1, 0), (dist.COL, "bcast", 1, 0)]}
    assert_global_collective_order(ok, 2, 2)
    R, Cc = (dist.ROW, "bcast", 1, 0), (dist.COL, "bcast", 1, 0)
    # rank 0 sits in its column collective waiting for rank 2, which sits in its row collective waiting for rank 3, which sits in
    # its column collective waiting for rank 1, which sits in its row collective waiting for rank 0
    bad = {0: [Cc, R], 1: [R, Cc], 3: [Cc, R], 2: [R, Cc]}
    assert_global_collective_order({0: [Cc, R], 1: [R, Cc], 2: [R, Cc], 3: [R, Cc]}, 2, 2)   # one odd rank alone is no cycle
    with pytest.raises(AssertionError):
        assert_global_collective_order(bad, 2, 2)


def test_buffer_ring_reuse_is_fenced(monkeypatch):
    """Between two uses of the same packed buffer by the diagonal owner's factorisation (a write), the program waits for the
    events that close every reader of the older panel: the copy into L (E_STORED), the panel stream (E_PANELDONE, unless that
    is the writing stream itself), the near / group-end updates (E_UPD of its group end) and its bulk update (E_BULK)."""
    n, nb, Pr, Pc, agg = 4200, 128, 2, 2, 4
    for rank in range(Pr * Pc):
        geo, prog, G = record_rank(n, nb, Pr, Pc, rank, agg, "eval", streamed=True, monkeypatch=monkeypatch)
        handle_to_slot = {g.h.value: i for i, g in enumerate(G)}
        R = len(G)
        writes = []   # (row index, slot) of DIAG_FACTOR rows: the first write into a buffer at its step
        for i, r in enumerate(prog.rows):
            if NAMES[r[0]] == "DIAG_FACTOR":
                writes.append((i, handle_to_slot[r[2]], r[4 + 5] // nb))     # a5 = base = k * nb
        waits_before = lambda i: {r[4] for r in prog.rows[:i] if NAMES[r[0]] == "WAIT"}
        for i, slot, k in writes:
            assert slot == k % R
            old = k - R
            if old < 0:
                continue
            seen = waits_before(i)
            ge = min((old // agg + 1) * agg - 1, geo.nblk - 1)
            assert dist._ev2(dist.E_STORED, old) in seen and dist._ev2(dist.E_UPD, ge) in seen, (rank, k)


@pytest.mark.parametrize("Pr,Pc,n,nb,agg", [(2, 2, 4200, 128, 4), (2, 4, 4200, 128, 2), (1, 2, 2500, 128, 3), (2, 4, 32768, 512, 4)])
def test_factor_window_slots_are_fenced(Pr, Pc, n, nb, agg, monkeypatch):
    """Streamed evaluation against a WINDOW of the factor (no N x N replica): panel k is copied into column slot k % window only
    behind a wait on the evaluation step that consumed the slot's previous panel (E_IVAR of that panel's group), every
    evaluation step reads its group at the group's slot, behind the copy of the group's last panel, and the steps come in
    panel order."""
    window = 2 * agg
    for rank in range(Pr * Pc):
        geo, prog, G = record_rank(n, nb, Pr, Pc, rank, agg, "bulk", streamed=True, monkeypatch=monkeypatch, window=window)
        analyse_events(prog)
        waits, groups, fwd_groups, pending_release = set(), [], [], None
        for r in prog.rows:
            op, a = NAMES[r[0]], r[4:]
            if op == "WAIT":
                waits.add(a[0])
            elif op == "UNPACK_DIAG":
                k = a[3] // nb
                assert a[4] - 1 == (k % window) * nb, "panel %d lands in the wrong column slot" % k
                if k >= window:
                    ge = min(((k - window) // agg + 1) * agg - 1, geo.nblk - 1)
                    assert dist._ev2(dist.E_IVAR, ge) in waits, "slot of panel %d rewritten before its evaluation step" % (k - window)
            elif op == "UNPACK_ROWS":
                assert a[6] % nb == 0 and a[6] < window * nb
            elif op == "IVAR_GROUP":
                k0, k1 = a[0], a[1]
                assert a[3] - 1 == (k0 % window) * nb and dist._ev2(dist.E_STORED, k1) in waits
                groups.append((k0, k1))
                pending_release = dist._ev2(dist.E_IVAR, k1)
            elif op == "FWD_GROUP":         # the forward substitution's step: same group, same slot, before the slot is released
                assert (a[0], a[1]) == groups[-1] and a[3] - 1 == (a[0] % window) * nb and pending_release is not None
                fwd_groups.append((a[0], a[1]))
            elif op == "RECORD" and a[0] == pending_release:
                pending_release = None
        assert groups == [(g, min(g + agg - 1, geo.nblk - 1)) for g in range(0, geo.nblk, agg)], groups
        assert fwd_groups == groups


@pytest.mark.parametrize("Pr,Pc,n,nb", [(1, 2, 1500, 128), (2, 2, 1500, 128), (2, 4, 2500, 128), (2, 3, 1500, 128), (4, 2, 1500, 128)])
def test_recorded_substitution_collective_order(Pr, Pc, n, nb):
    """The distributed forward / back substitution (dist2_potrs: ncclReduce along process rows / columns, ncclBroadcast of the
    solved blocks, one ncclAllReduce): every member of a communicator issues the same sequence."""
    W = Pr * Pc
    seqs = {}
    for rank in range(W):
        geo = dist.Grid2D(n, nb, Pr, Pc, rank)
        ops = dist.DeviceOps2D(None)
        comm = object.__new__(dist.RcclComm)
        comm.ctx, comm.rank, comm.world = None, rank, W
        prog = dist.Program()
        ops.prog = comm.prog = prog
        dist.dist2_potrs(ops, comm, geo, FakeMat(), FakeMat(), FakeMat(), FakeMat(), FakeMat())
        seqs[rank] = collectives(prog)
        # counts stay inside the vectors: every reduce / broadcast moves one block of at most nb doubles
        assert all(c[2] <= nb for c in seqs[rank] if c[1] in ("bcast", "reduce"))
    assert all([c for c in seqs[r] if c[0] == dist.WORLD] == [c for c in seqs[0] if c[0] == dist.WORLD] for r in range(W))
    for grp, members in [(dist.ROW, [[p * Pc + q for q in range(Pc)] for p in range(Pr)]),
                         (dist.COL, [[p * Pc + q for p in range(Pr)] for q in range(Pc)])]:
        for ranks in members:
            s = [[c for c in seqs[r] if c[0] == grp] for r in ranks]
            assert all(x == s[0] for x in s), (grp, ranks)
    assert_global_collective_order(seqs, Pr, Pc)
