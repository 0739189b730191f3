"""ReplayComm: a MEASUREMENT double of the communicator interface of gpexp_amd.dist (RcclComm), kept out of the product package
(VERDICT r3 weak 9).  Used by scripts/dist_replay.py, scripts/graph_capture_bisect.py and bench.py's `multi_gpu_replay` object."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from gpexp_amd.dist import Emitter, OP, ROW, COL  # noqa: E402
from gpexp_amd._lib import as_f64  # noqa: E402


class ReplayComm(Emitter):
    """ONE process plays rank `rank` of a `world`-rank grid on one GPU: every receive of the panel loop becomes a device copy
    of the same bytes out of the complete factor `Lref` resident on this GPU (gpx_dist2_pack_*), sends cost nothing (the
    sender's data is its own), and nothing waits for a peer.  What it measures: the rank's kernel sequence, its GPU time per
    strand and the host issue time -- not xGMI.  Only the factorisation + streamed evaluation are replayable (the substitution
    sweeps need the peers' partial sums)."""
    recordable = True

    def __init__(self, ctx, world, rank, Lref, pace_us=None):
        self.ctx, self.world, self.rank, self.Lref = ctx, int(world), int(rank), Lref
        self.k = None
        self.bytes_in = 0     # bytes the collectives would have delivered to this rank (per recording)
        # PACED replay: pace_us[k] = microseconds the panel of step k takes to arrive after the panel of step k-1 when ANOTHER
        # process column solves it (this rank's own holder steps take what they take).  Without pacing every foreign panel
        # arrives at once, the rank is never idle, and its busy time says nothing about the chain ACROSS ranks -- step k+1's
        # panel solve needs step k's panel, so a real run's factorisation time is the SUM of the holders' per-step latencies.
        # With pacing set to the latencies this rank shows in its own holder steps (scripts/dist_replay.py iterates to a fixed
        # point), the replayed step reproduces that chain on one GPU.  xGMI transfer time is still not in it.
        # pace_us may also be a dict(panel=..., dfact=..., early=...) of per-step arrays (us after the arrival of panel k-1 at
        # which the PRODUCER had panel k delivered / diagonal block k factored / block row k+1 of panel k solved): the two small
        # broadcasts of the DIAGONAL chain -- L_kk down its process column, L[k+1, k] along its process row -- are then held back
        # too, each until "arrival of panel k-1 (a wall-clock stamp on this stream) + the producer's figure" (gpx_dbg_spin_until).
        # With the panels alone paced they arrive the moment this rank asks, and a replayed owner factors its diagonal block
        # earlier than any real grid could hand it the block row it needs.
        self.pace = pace_us if isinstance(pace_us, dict) else None
        self.pace_us = None if self.pace is not None else pace_us

    def set_grid(self, Pr, Pc):
        assert Pr * Pc == self.world
        self.grid = (Pr, Pc)

    CH0 = 512   # stamp slots CH0 + k + 1: chunk 0 of panel k has arrived (slot k + 1: all of panel k); gpx_debug.h has 1024 slots

    def at_step(self, geo, k):
        self.geo, self.k = geo, k
        self.chunk, self.last_chunk = 0, True
        if k == 0 and self.pace is not None:
            self._emit(OP["SPIN"], (), (0, 2))           # stamp slot 0: the step starts (slot k + 1: panel k has arrived)
            self._emit(OP["SPIN"], (), (self.CH0, 2))

    def at_chunk(self, c, last):
        """round 5: the panel loop announces which row chunk of panel k the next panel_bcast delivers, and whether it is the last"""
        self.chunk, self.last_chunk = int(c), bool(last)

    def _hold(self, kind, base=0):
        """foreign delivery of step k: not before the stamp of panel k-1's arrival (chunk 0: of ITS chunk 0's) + the producer's figure"""
        if self.pace is None or self.pace.get(kind) is None:
            return
        us = int(self.pace[kind][self.k])
        if us > 0:
            self._emit(OP["SPIN"], (), (base + self.k, 3, us))

    def _rows(self, buf, off, m, first_block, stride, k):
        geo = self.geo
        self._emit(OP["PACK_ROWS"], (self.Lref, buf), (first_block, stride, k * geo.nb, off, m, geo.height(k), geo.nb))
        self.bytes_in += 8 * m * geo.gld

    def _diag(self, buf, off, k):
        geo = self.geo
        self._emit(OP["PACK_DIAG"], (self.Lref, buf), (k * geo.nb, geo.height(k), geo.nb, off))
        self.bytes_in += 8 * geo.dsz

    def bcast_grp(self, buf, offset, count, root, grp):
        geo, k = self.geo, self.k
        mine = geo.pc if grp == ROW else geo.pr
        if count == 0 or mine == root:
            return
        if grp == COL:
            assert offset == geo.piece_off(k % geo.Pr) and count == geo.dsz
            self._hold("dfact")
            self._diag(buf, offset, k)
        else:
            assert grp == ROW and count == geo.height(k + 1) * geo.gld
            self._hold("early")
            self._rows(buf, offset, geo.height(k + 1), k + 1, 1, k)

    def bcast_grp2(self, sbuf, soff, rbuf, roff, count, root, grp):
        """the early block row into its own buffer (row communicator only)"""
        geo, k = self.geo, self.k
        assert grp == ROW and count == geo.height(k + 1) * geo.gld
        if count == 0:
            return
        if geo.pc != root:
            self._hold("early")
        self._rows(rbuf, roff, geo.height(k + 1), k + 1, 1, k)      # (the root keeps a copy too, as ncclBroadcast gives it)

    def panel_bcast(self, buf, pieces):
        geo, k = self.geo, self.k
        foreign = geo.pc != k % geo.Pc
        if self.pace_us is not None and foreign and self.chunk == 0 and int(self.pace_us[k]) > 0:
            self._emit(OP["SPIN"], (), (int(self.pace_us[k]), 1))      # the foreign holder's latency (COMM stream, in order)
        if foreign:
            if self.chunk == 0:
                self._hold("panel0", self.CH0)
            if self.last_chunk:
                self._hold("panel")
        for off, cnt, root in pieces:
            if root == self.rank or cnt == 0:
                continue
            p = root // geo.Pc
            rel = off - geo.piece_off(p)
            if rel == 0:                                   # the diagonal region leads the piece of the holder's process row
                assert p == k % geo.Pr and cnt >= geo.dsz and (cnt - geo.dsz) % geo.gld == 0
                self._diag(buf, off, k)
                off, cnt, rel = off + geo.dsz, cnt - geo.dsz, geo.dsz
            assert rel >= geo.dsz and (rel - geo.dsz) % geo.gld == 0 and cnt % geo.gld == 0
            r0, m = (rel - geo.dsz) // geo.gld, cnt // geo.gld   # rows [r0, r0 + m) of piece p: a whole number of blocks above them
            assert r0 % geo.nb == 0 and r0 + m <= geo.piece_rows(p, k)
            if m > 0:
                self._rows(buf, off, m, p + (geo.li0(p, k) + r0 // geo.nb) * geo.Pr, geo.Pr, k)
        if self.pace is not None:
            if self.chunk == 0:
                self._emit(OP["SPIN"], (), (self.CH0 + k + 1, 2))   # stamp: chunk 0 of panel k has arrived
            if self.last_chunk:
                self._emit(OP["SPIN"], (), (k + 1, 2))              # stamp: panel k has arrived

    def reduce_grp(self, *a):
        raise NotImplementedError("the substitution sweeps are not replayable on one rank")

    allreduce = reduce_grp

    def allgather(self, vec):
        return np.tile(as_f64(np.atleast_1d(vec)), (self.world, 1))

    def barrier(self):
        self.ctx.sync()

    def max_float(self, v):
        return float(v)

    def close(self):
        pass
