"""Worker for the multi-rank tests; launched as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node W --master-addr 127.0.0.1 --master-port P \
        tests/dist_worker.py --mode {cpu,gpu} --npts N --blk NB
cpu: the real panel loop (gpexp_amd.dist.dist_potrf) over gloo with a NumPy stand-in for the device primitives
     (test double, built on the oracle's kernel functions) -- checks ownership / panel / broadcast logic.
gpu: the real HIP primitives; ranks share GPU 0 and exchange panels through the host-staged gloo communicator
     (RCCL refuses two ranks on one device); rank 0 compares against the single-GPU path.
Prints "DIST_OK ..." on success from rank 0.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the workers below drive the runners directly and use the single-GPU class API as their yardstick: the class API must not
# attach itself to the process group behind their back (the cpu-api / gpu-api modes attach explicitly)
os.environ.setdefault("GPX_DIST_ATTACH", "0")

from gpexp_amd import dist  # noqa: E402
import dist_testcomm  # noqa: E402  (tests/: the gloo communicators)


class NumpyMat:
    def __init__(self, a):
        self.a = a


class NumpyOps:
    """NumPy double of gpexp_amd.dist.DeviceOps (same call signatures, same panel-buffer layout)."""

    def __init__(self, spec_dict):
        from oracle import gpexp_oracle as orc
        self.orc = orc
        self.spec = spec_dict
        self.aux = {}

    def alloc_matrix(self, n):
        np_ = dist.padded(n)
        return NumpyMat(np.full((np_, np_), np.nan))  # NaN: reading a block nobody wrote must show up

    def alloc_panel(self, n, nb):
        return NumpyMat(np.zeros(dist.panel_elems(n, nb)))

    def kfill_owned(self, spec, X, K, nugget, nb, rank, world):
        n = X.shape[0]
        np_ = K.a.shape[0]
        full = np.eye(np_)
        full[:n, :n] = self.orc.cov_matrix(self.spec, X, float(nugget), row_loop=False)
        for j in dist.owned_blocks(n, nb, rank, world):
            c0, c1 = j * nb, min((j + 1) * nb, np_)
            K.a[c0:, c0:c1] = full[c0:, c0:c1]

    def begin(self):
        self._info = 0

    def info(self):
        return self._info

    def stream(self, which):  # the double executes synchronously: streams / events are no-ops
        pass

    def record(self, ev):
        pass

    def wait(self, ev):
        pass

    def panel_factor(self, K, k, nb, P):
        np_ = K.a.shape[0]
        r0 = k * nb
        w = min(nb, np_ - r0)
        rows = np_ - r0
        pan = K.a[r0:, r0:r0 + w].copy()
        assert not np.isnan(np.tril(pan[:w])).any(), "owner factors a panel it never assembled/updated"
        try:
            L11 = np.linalg.cholesky(np.tril(pan[:w]) + np.tril(pan[:w], -1).T)
        except np.linalg.LinAlgError:
            if self._info == 0:
                self._info = r0 + 1
            L11 = np.eye(w)
        pan[:w] = L11
        if rows > w:
            pan[w:] = np.linalg.solve(L11, pan[w:].T).T
        buf = np.zeros((rows, nb))
        buf[:, :w] = pan
        P.a[:rows * nb] = buf.ravel()
        for q in range(w // 128):
            blk = L11[q * 128:(q + 1) * 128, q * 128:(q + 1) * 128]
            P.a[rows * nb + q * 128 * 128: rows * nb + (q + 1) * 128 * 128] = np.linalg.inv(blk).ravel()

    def _panel(self, K, k, nb, P):
        np_ = K.a.shape[0]
        r0 = k * nb
        w = min(nb, np_ - r0)
        rows = np_ - r0
        return np_, r0, w, rows, P.a[:rows * nb].reshape(rows, nb)[:, :w]

    def panel_store(self, K, k, nb, P):
        np_, r0, w, rows, pan = self._panel(K, k, nb, P)
        K.a[r0:, r0:r0 + w] = pan
        self.aux[k] = P.a[rows * nb: rows * nb + (w // 128) * 128 * 128].copy()

    def panel_update(self, K, k, nb, P, j0, j1, rank, world):
        np_, r0, w, rows, pan = self._panel(K, k, nb, P)
        nblk = (np_ + nb - 1) // nb
        for j in range(max(j0, k + 1), min(j1, nblk)):
            if j % world != rank:
                continue
            c0, c1 = j * nb, min((j + 1) * nb, np_)
            K.a[c0:, c0:c1] -= pan[c0 - r0:] @ pan[c0 - r0:c1 - r0].T

    def finish(self, K):
        pass

    # streamed evaluation (same semantics as gpx_dist_ivar_step): one right-looking solve step per stored panel
    def alloc_cross(self, n, m):
        return NumpyMat(np.zeros((dist.padded(n), m)))

    def cross_fill(self, spec, X, Z, B):
        B.a[:] = 0.0
        B.a[:X.shape[0], :] = self.orc.cross_matrix(self.spec, Z, X).T

    def ivar_step(self, K, k, nb, B, c0=None):
        np_ = K.a.shape[0]
        r0 = k * nb
        c0 = r0 if c0 is None else c0
        w = min(nb, np_ - r0)
        Lkk = np.tril(K.a[r0:r0 + w, c0:c0 + w])
        assert not np.isnan(Lkk).any() and not np.isnan(K.a[r0 + w:, c0:c0 + w]).any(), "evaluation reads a panel that is not stored"
        B.a[r0:r0 + w] = np.linalg.solve(Lkk, B.a[r0:r0 + w])
        if r0 + w < np_:
            B.a[r0 + w:] -= K.a[r0 + w:, c0:c0 + w] @ B.a[r0:r0 + w]

    def ivar_group(self, K, k0, k1, nb, B, c0=None):
        for k in range(k0, k1 + 1):
            self.ivar_step(K, k, nb, B, None if c0 is None else c0 + (k - k0) * nb)

    def variances(self, spec, Z, B, n):
        return self.orc.kernel_diag(self.spec, Z) - np.sum(B.a[:n] ** 2, axis=0)


class NumpyOps2D(NumpyOps):
    """NumPy double of gpexp_amd.dist.DeviceOps2D: same call signatures, same packed-buffer layout, executed
    synchronously -- checks the index logic of the 2-D block-cyclic loop (ownership, piece offsets, update operands,
    unpacking into the replicated factor, the distributed substitution sweeps)."""
    poison = True   # unwritten storage is NaN and every read of it asserts; switched off for the non-PD run (inf - inf)

    def alloc_local(self, geo):
        return NumpyMat(np.full((max(geo.local_rows(geo.pr), 1), max(geo.local_cols(geo.pc), 1)), np.nan))

    def alloc_buf(self, geo):
        return NumpyMat(np.full(geo.buf_elems(), np.nan))

    def alloc_vec(self, n):
        return NumpyMat(np.zeros(max(int(n), 1)))

    def points(self, x):
        return np.asarray(x, dtype=float)

    def kfill_local(self, spec, X, A, nugget, geo):
        n, nb = X.shape[0], geo.nb
        full = np.eye(geo.np)
        full[:n, :n] = self.orc.cov_matrix(self.spec, X, float(nugget), row_loop=False)
        A.a[:] = np.nan
        for I in range(geo.pr, geo.nblk, geo.Pr):
            for J in range(geo.pc, geo.nblk, geo.Pc):
                if I >= J:
                    A.a[(I // geo.Pr) * nb:(I // geo.Pr) * nb + geo.height(I),
                        (J // geo.Pc) * nb:(J // geo.Pc) * nb + geo.height(J)] = \
                        full[I * nb:I * nb + geo.height(I), J * nb:J * nb + geo.height(J)]

    @staticmethod
    def _D(G, doff, nb):
        return G.a[doff:doff + nb * nb].reshape(nb, nb)

    def diag_factor(self, A, lr, lc, w, G, doff, nb, base, n_valid):
        self._inside(A, lr, w, lc, w)
        blk = A.a[lr:lr + w, lc:lc + w]
        assert not self.poison or not np.isnan(np.tril(blk)).any(), "owner factors a diagonal block it never assembled/updated"
        try:
            L11 = np.linalg.cholesky(np.tril(blk) + np.tril(blk, -1).T)
        except (np.linalg.LinAlgError, ValueError):
            if self._info == 0:
                self._info = base + 1
            L11 = np.eye(w)
        D = self._D(G, doff, nb)
        D[:] = 0.0
        D[:w, :w] = L11
        A.a[lr:lr + w, lc:lc + w] = L11
        inv = G.a[doff + nb * nb: doff + nb * nb + (nb // 128) * 128 * 128]
        inv[:] = 0.0
        for q in range(w // 128):
            inv[q * 128 * 128:(q + 1) * 128 * 128] = np.linalg.inv(L11[q * 128:(q + 1) * 128, q * 128:(q + 1) * 128]).ravel()
        self.aux[("A", lr)] = inv[:(w // 128) * 128 * 128].copy()

    # the same in parts (dist2_potrf_enqueue, staged): the block is copied into the packed buffer ahead of its last update
    def diag_stage(self, A, lr, lc, w, G, doff, nb):
        self._inside(A, lr, w, lc, w)
        blk = A.a[lr:lr + w, lc:lc + w]
        assert not self.poison or not np.isnan(np.tril(blk)).any(), "owner stages a diagonal block it never assembled/updated"
        D = self._D(G, doff, nb)
        D[:] = np.nan if self.poison else 0.0
        D[:w, :w] = np.tril(blk) + np.tril(blk, -1).T

    def diag_update(self, G, doff, h, S, soff, w, nb):
        gld = nb + dist.G_SKEW
        assert soff + h * gld <= S.a.size
        rows = S.a[soff:soff + h * gld].reshape(h, gld)[:, :w]
        assert not self.poison or not np.isnan(rows).any(), "the staged diagonal block is updated by a block row that never arrived"
        self._D(G, doff, nb)[:h, :h] -= rows @ rows.T

    def diag_factor_staged(self, A, lr, lc, w, G, doff, nb, base, n_valid):
        D = self._D(G, doff, nb)
        blk = D[:w, :w].copy()
        assert not self.poison or not np.isnan(np.tril(blk)).any(), "owner factors a diagonal block that was never staged"
        try:
            L11 = np.linalg.cholesky(np.tril(blk) + np.tril(blk, -1).T)
        except (np.linalg.LinAlgError, ValueError):
            if self._info == 0:
                self._info = base + 1
            L11 = np.eye(w)
        D[:] = 0.0
        D[:w, :w] = L11
        inv = G.a[doff + nb * nb: doff + nb * nb + (nb // 128) * 128 * 128]
        inv[:] = 0.0
        for q in range(w // 128):
            inv[q * 128 * 128:(q + 1) * 128 * 128] = np.linalg.inv(L11[q * 128:(q + 1) * 128, q * 128:(q + 1) * 128]).ravel()

    def diag_store(self, A, lr, lc, w, G, doff, nb, dslot=None):
        self._inside(A, lr, w, lc, w)
        L11 = self._D(G, doff, nb)[:w, :w]
        assert not self.poison or not np.isnan(L11).any(), "the stored diagonal block was never factored"
        A.a[lr:lr + w, lc:lc + w] = L11
        self.aux[("A", lr)] = G.a[doff + nb * nb: doff + nb * nb + (w // 128) * 128 * 128].copy()

    @staticmethod
    def _inside(A, lr0, m, lc, w):
        """The C primitives reject blocks outside the local matrix (NumPy slicing would not)."""
        assert 0 <= lr0 and lr0 + m <= A.a.shape[0] and 0 <= lc and lc + w <= A.a.shape[1], (lr0, m, lc, w, A.a.shape)

    def panel_inv(self, G, doff, nb, w):
        assert not self.poison or not np.isnan(self._D(G, doff, nb)[:w, :w][np.tril_indices(w)]).any(), \
            "the inverse is built from a diagonal block that has not arrived"

    def panel_copyback(self, A, lr0, m, lc, w, G, roff, nb):
        pass      # (the double's panel_trsm writes both places at once)

    # the finished factor re-streamed out of the local matrix (dist2_restream_enqueue)
    def panel_pack(self, A, lr0, m, lc, w, G, roff, nb):
        self._inside(A, lr0, m, lc, w)
        rows = A.a[lr0:lr0 + m, lc:lc + w]
        assert not self.poison or not np.isnan(rows).any(), "a holder re-streams rows of a panel it never solved"
        gld = nb + dist.G_SKEW
        assert roff + m * gld <= G.a.size
        G.a[roff:roff + m * gld] = np.nan if self.poison else 0.0
        G.a[roff:roff + m * gld].reshape(m, gld)[:, :w] = rows

    def diag_pack(self, A, lr, lc, w, G, doff, nb):
        self._inside(A, lr, w, lc, w)
        blk = np.tril(A.a[lr:lr + w, lc:lc + w])
        assert not self.poison or not np.isnan(blk).any(), "the owner re-streams a diagonal block it never factored"
        D = self._D(G, doff, nb)
        D[:] = np.nan if self.poison else 0.0
        D[:w, :w] = blk
        G.a[doff + nb * nb: doff + nb * nb + (w // 128) * 128 * 128] = self.aux.get(("A", lr), 0.0)

    def panel_trsm(self, A, lr0, m, lc, w, G, doff, roff, nb, dslot=None, prepared=False, copy_back=True):
        self._inside(A, lr0, m, lc, w)
        if m == 0:
            return
        L11 = self._D(G, doff, nb)[:w, :w]
        assert not self.poison or not np.isnan(L11).any(), "holder solves against a diagonal block that never arrived"
        X = A.a[lr0:lr0 + m, lc:lc + w]
        assert not self.poison or not np.isnan(X).any(), "holder solves rows it never assembled/updated"
        X[:] = np.linalg.solve(L11, X.T).T
        gld = nb + dist.G_SKEW
        rows = G.a[roff:roff + m * gld].reshape(m, gld)[:, :nb]   # the 16 pad doubles of a row are never read or written
        rows[:] = 0.0
        rows[:, :w] = X

    def update(self, A, lr0, m, lc0, n, G, aoff, boff, w, nb):
        self._inside(A, lr0, m, lc0, n)
        gld = nb + dist.G_SKEW
        assert aoff + m * gld <= G.a.size and boff + n * gld <= G.a.size
        if m == 0 or n == 0:
            return
        a = G.a[aoff:aoff + m * gld].reshape(m, gld)[:, :w]
        b = G.a[boff:boff + n * gld].reshape(n, gld)[:, :w]
        assert not self.poison or not (np.isnan(a).any() or np.isnan(b).any()), "update reads a piece that never arrived"
        A.a[lr0:lr0 + m, lc0:lc0 + n] -= a @ b.T

    def update_multi(self, A, lr0, m, lc0, n, geo, Gs, ks, below_diag):
        """Same semantics as gpx_dist2_update_multi: every local block (li, lj) of the range with global I > J (I >= J
        unless below_diag) gets minus the sum over the panels ks of  rows_I(panel) rows_J(panel)^T, the operands taken
        from the packed buffers exactly where the device kernel reads them."""
        self._inside(A, lr0, m, lc0, n)
        nb, Pr, Pc, pr, pc = geo.nb, geo.Pr, geo.Pc, geo.pr, geo.pc
        assert lr0 % nb == 0 and lc0 % nb == 0 and len(Gs) == len(ks) <= 8 and Pr <= 4
        for lj in range(lc0 // nb, (lc0 + n + nb - 1) // nb):
            J = lj * Pc + pc
            cw = min(nb, lc0 + n - lj * nb)
            for li in range(lr0 // nb, (lr0 + m + nb - 1) // nb):
                I = li * Pr + pr
                if I < J or (below_diag and I == J):
                    continue
                rh = min(nb, lr0 + m - li * nb)
                acc = np.zeros((rh, cw))
                for G, k in zip(Gs, ks):
                    assert I > k and J > k, "update touches blocks that are not behind the panel"
                    gld = geo.gld
                    aoff = geo.piece_off(pr) + geo.dsz + (li - geo.li0(pr, k)) * nb * gld
                    pj = J % Pr
                    boff = geo.piece_off(pj) + geo.dsz + (J // Pr - geo.li0(pj, k)) * nb * gld
                    assert aoff + rh * gld <= geo.piece_off(pr) + geo.piece_stride
                    assert boff + cw * gld <= geo.piece_off(pj) + geo.piece_stride
                    a = G.a[aoff:aoff + rh * gld].reshape(rh, gld)[:, :nb]
                    b = G.a[boff:boff + cw * gld].reshape(cw, gld)[:, :nb]
                    assert not self.poison or not (np.isnan(a).any() or np.isnan(b).any()), \
                        "update reads a piece that never arrived (panel %d, block %d,%d)" % (k, I, J)
                    acc += a @ b.T
                blk = A.a[li * nb:li * nb + rh, lj * nb:lj * nb + cw]
                assert not self.poison or I == J or not np.isnan(blk).any(), "update of a block that was never assembled"
                blk -= acc

    def spin(self, ms):
        pass

    def unpack_rows(self, G, roff, m, w, nb, L, first_block, stride, col0):
        gld = nb + dist.G_SKEW
        rows = G.a[roff:roff + m * gld].reshape(m, gld)[:, :w]
        for t in range((m + nb - 1) // nb):
            h = min(nb, m - t * nb)
            g0 = (first_block + t * stride) * nb
            L.a[g0:g0 + h, col0:col0 + w] = rows[t * nb:t * nb + h]

    def unpack_diag(self, G, doff, w, nb, L, r0, c0=None):
        if c0 is None:
            c0 = r0
        else:
            L.a[:, c0:c0 + w] = np.nan       # window of block columns: the slot's previous panel is gone
        L.a[r0:r0 + w, c0:c0 + w] = self._D(G, doff, nb)[:w, :w]

    def alloc_window(self, n, cols):
        return NumpyMat(np.full((dist.padded(n), cols), np.nan))

    def fwd_group(self, K, k0, k1, nb, v, c0=None):
        np_ = K.a.shape[0]
        y = v.a.reshape(-1)
        for k in range(k0, k1 + 1):
            r0 = k * nb
            c = r0 if c0 is None else c0 + (k - k0) * nb
            w = min(nb, np_ - r0)
            blk = K.a[r0:, c:c + w]
            assert not np.isnan(np.tril(blk[:w])).any() and not np.isnan(blk[w:]).any(), "forward step reads a panel that is not stored"
            y[r0:r0 + w] = np.linalg.solve(np.tril(blk[:w]), y[r0:r0 + w])
            y[r0 + w:] -= blk[w:] @ y[r0:r0 + w]

    def replica_solve(self, L, y0, alpha):
        Lt = np.tril(L.a)
        assert not np.isnan(Lt).any(), "the replicated factor is incomplete"
        alpha.a.reshape(-1)[:] = np.linalg.solve(Lt.T, np.linalg.solve(Lt, y0.a.reshape(-1)))

    def replica_logdet(self, L):
        return 2.0 * float(np.sum(np.log(np.diag(L.a))))      # (the padding of the matrix is an identity block)

    def trsv_diag(self, A, lr, lc, w, v, voff, transposed):
        self._inside(A, lr, w, lc, w)
        Lkk = np.tril(A.a[lr:lr + w, lc:lc + w])
        v.a[voff:voff + w] = np.linalg.solve(Lkk.T if transposed else Lkk, v.a[voff:voff + w])

    def gemv(self, A, lr0, m, lc, w, x, xoff, acc, aoff, transposed):
        self._inside(A, lr0, m, lc, w)
        if m == 0 or w == 0:
            return
        blk = A.a[lr0:lr0 + m, lc:lc + w]
        assert not np.isnan(blk).any(), "substitution reads a block of the factor that was never written"
        if transposed:
            acc.a[aoff:aoff + w] -= blk.T @ x.a[xoff:xoff + m]
        else:
            acc.a[aoff:aoff + m] -= blk @ x.a[xoff:xoff + w]

    def logdet_acc(self, A, lr, lc, w, n_valid, acc):
        acc.a[0] += 2.0 * np.sum(np.log(np.diag(A.a[lr:lr + w, lc:lc + w])[:n_valid]))

    def vec_op(self, dst, doff, src, soff, n, mode):
        if mode == 2:
            dst.a[doff:doff + n] = 0.0
        elif mode == 1:
            dst.a[doff:doff + n] += src.a[soff:soff + n]
        else:
            dst.a[doff:doff + n] = src.a[soff:soff + n]

    def vec_to_host(self, v, n):
        return v.a[:n].copy()

    def vec_from_host(self, v, a):
        v.a[:len(a)] = a

    def alloc_matrix(self, n):
        np_ = dist.padded(n)
        return NumpyMat(np.full((np_, np_), np.nan))

    def finish(self, L):
        pass

    def posterior_var(self, spec, L, X, Z):
        n = X.shape[0]
        W = np.linalg.solve(np.tril(L.a[:n, :n]), self.orc.cross_matrix(self.spec, Z, X).T)
        return self.orc.kernel_diag(self.spec, Z) - np.sum(W ** 2, axis=0)


class NumpyComm:
    def __init__(self):
        self.group = dist_testcomm._TorchGroup()
        self.rank, self.world = self.group.rank, self.group.world

    # 2-D path: same interface as gpexp_amd.dist.RcclComm, on host arrays over gloo
    def set_grid(self, Pr, Pc):
        if getattr(self, "grid", None) != (Pr, Pc):
            self.group.make_grid(Pr, Pc)
            self.grid = (Pr, Pc)

    def bcast_grp(self, buf, offset, count, root, grp):
        seg = np.ascontiguousarray(buf.a[offset:offset + count])
        buf.a[offset:offset + count] = self.group.bcast_array_grp(seg, root, grp)

    def bcast_grp2(self, sbuf, soff, rbuf, roff, count, root, grp):
        seg = np.ascontiguousarray(sbuf.a[soff:soff + count])
        rbuf.a[roff:roff + count] = self.group.bcast_array_grp(seg, root, grp)

    def reduce_grp(self, buf, offset, count, root, grp):
        seg = np.ascontiguousarray(buf.a[offset:offset + count])
        buf.a[offset:offset + count] = self.group.reduce_array_grp(seg, root, grp)

    def allreduce(self, buf, offset, count):
        seg = np.ascontiguousarray(buf.a[offset:offset + count])
        buf.a[offset:offset + count] = self.group.allreduce_array(seg)

    def allreduce_host(self, vec):
        return self.group.allreduce_array(np.array(np.atleast_1d(vec), dtype=np.float64))

    def panel_bcast(self, buf, pieces):
        for off, cnt, root in pieces:
            seg = np.ascontiguousarray(buf.a[off:off + cnt])
            self.group.bcast_array(seg, root)
            buf.a[off:off + cnt] = seg

    def bcast_panel(self, P, count, root):
        buf = np.ascontiguousarray(P.a[:count])
        self.group.bcast_array(buf, root)
        P.a[:count] = buf

    def allgather(self, vec):
        return self.group.allgather(np.atleast_1d(vec))

    def barrier(self):
        self.group.barrier()

    def max_float(self, v):
        return self.group.max_float(v)


def run_cpu(args):
    rng = np.random.default_rng(args.n)
    d = 3
    X = rng.uniform(-1, 1, (args.n, d))
    spec = dict(kind="se", cl=[0.3], signalSize=1.0, d=d)
    comm = NumpyComm()
    ops = NumpyOps(spec)
    K = ops.alloc_matrix(args.n)
    P = [ops.alloc_panel(args.n, args.nb), ops.alloc_panel(args.n, args.nb)]
    ops.kfill_owned(None, X, K, 0.05, args.nb, comm.rank, comm.world)
    info = dist.dist_potrf(ops, comm, K, args.n, args.nb, P)
    assert info == 0
    from oracle import gpexp_oracle as orc
    Lref = np.linalg.cholesky(orc.cov_matrix(spec, X, 0.05, row_loop=False))
    L = np.tril(K.a[:args.n, :args.n])
    err = np.max(np.abs(L - Lref)) / np.max(np.abs(Lref))
    assert err < 1e-12, err
    # every rank must hold the complete factor (that is what lets evaluation shard without moving L)
    allerr = comm.allgather(np.array([err]))[:, 0]
    # streamed evaluation hook: one solve step per stored panel, in panel order, ends with L^-1 K(X, Z_local)
    Z = rng.uniform(-1, 1, (41 + comm.world, d))
    lo, hi = dist.eval_slice(len(Z), comm.rank, comm.world)
    K2 = ops.alloc_matrix(args.n)
    ops.kfill_owned(None, X, K2, 0.05, args.nb, comm.rank, comm.world)
    B = ops.alloc_cross(args.n, hi - lo)
    ops.cross_fill(None, X, Z[lo:hi], B)
    seen = []

    def hook(k):
        seen.append(k)
        ops.ivar_step(K2, k, args.nb, B)

    assert dist.dist_potrf(ops, comm, K2, args.n, args.nb, P, on_stored=hook) == 0
    assert seen == list(range(dist.num_blocks(args.n, args.nb)))
    Wref = np.linalg.solve(Lref, orc.cross_matrix(spec, Z[lo:hi], X).T)
    assert np.max(np.abs(B.a[:args.n] - Wref)) <= 1e-11 * max(np.max(np.abs(Wref)), 1e-300)
    var = ops.variances(None, Z[lo:hi], B, args.n)
    allvar = np.concatenate([v[:c] for v, c in zip(comm.allgather(np.pad(var, (0, 64 - len(var)))), [e - b for b, e in (dist.eval_slice(len(Z), r, comm.world) for r in range(comm.world))])])
    m_ref = orc.fit(spec, X, np.zeros(args.n), 0.05)
    assert np.max(np.abs(allvar - orc.posterior(spec, m_ref, Z, compvar=1)[1])) <= 1e-10
    # evaluation slices tile the index range exactly
    m = 1000 + comm.world
    sl = [dist.eval_slice(m, r, comm.world) for r in range(comm.world)]
    assert sl[0][0] == 0 and sl[-1][1] == m and all(sl[i][1] == sl[i + 1][0] for i in range(comm.world - 1))
    # non-PD detection is agreed on by all ranks
    Kb = ops.alloc_matrix(args.n)
    ops.kfill_owned(None, X, Kb, 0.05, args.nb, comm.rank, comm.world)
    if dist.owner(0, comm.world) == comm.rank:
        Kb.a[5, 5] = -1.0
    info = dist.dist_potrf(ops, comm, Kb, args.n, args.nb, P)
    assert info == 1, info
    t = comm.max_float(float(comm.rank))
    assert t == comm.world - 1
    if comm.rank == 0:
        print("DIST_OK cpu world=%d n=%d nb=%d maxerr=%.2e" % (comm.world, args.n, args.nb, float(allerr.max())), flush=True)


def run_cpu2d(args):
    """The real 2-D block-cyclic loop (dist2_potrf / dist2_potrs / dist2_logdet / DistFitIvar2D) over gloo with the NumPy
    double: factor (distributed AND replicated copies), non-PD agreement, streamed hook order, alpha, logdet, IVAR."""
    from oracle import gpexp_oracle as orc
    rng = np.random.default_rng(args.n)
    d = 3
    n, nb = args.n, args.nb
    X = rng.uniform(-1, 1, (n, d))
    y = np.sin(2 * np.pi * X.sum(1) / d) + 0.2 * rng.standard_normal(n)
    # --mpts below 10: FEWER evaluation points than ranks may have -- some ranks stream nothing, keep no window, and the
    # forward substitution cannot ride along (every rank must take that decision alike)
    mz = args.m if args.m < 10 else 53 + 2 * int(os.environ.get("WORLD_SIZE", "1"))
    Z = rng.uniform(-1, 1, (mz, d))
    spec = dict(kind="se", cl=[0.3], signalSize=1.0, d=d)
    comm = NumpyComm()
    grid = tuple(int(v) for v in args.grid.split("x")) if args.grid else dist.choose_grid(comm.world)
    ops = NumpyOps2D(spec)
    Kref = orc.cov_matrix(spec, X, 0.05, row_loop=False)
    Lref = np.linalg.cholesky(Kref)
    aref = np.linalg.solve(Kref, y)
    # evaluation after the fit against the replica / streamed against a WINDOW of the factor (no replica) / streamed + replica
    for streamed, replicate in ((False, None), (True, None), (True, True)):
        run = dist.DistFitIvar2D(None, comm, None, X, y, Z, 0.05, nb=nb, ops=ops, streamed=streamed, grid=grid,
                                 replicate=replicate)
        geo = run.geo
        assert (geo.Pr, geo.Pc) == grid and geo.Pr * geo.Pc == comm.world
        assert bool(run.window) == (streamed and not replicate)
        assert run.fused_fwd == (bool(run.window) and mz >= comm.world)
        if run.window:
            assert run.L.a.shape == (geo.np, 2 * run.agg * nb), run.L.a.shape     # no N x N copy of the factor on this rank
        seen = []
        if streamed:
            orig = ops.ivar_group

            def spy(K, k0, k1, nb_, B, c0=None):
                assert (c0 is None) == (not run.window) and (c0 is None or c0 == (k0 % run.window) * nb_)
                seen.extend(range(k0, k1 + 1))
                orig(K, k0, k1, nb_, B, c0)
            ops.ivar_group = spy
        ll, iv = run.step()
        if streamed:
            ops.ivar_group = orig
            if run.B is not None:
                assert seen == list(range(geo.nblk)), seen
        if not run.window:        # replicated factor on every rank
            L = np.tril(run.L.a[:n, :n])
            err = np.max(np.abs(L - Lref)) / np.max(np.abs(Lref))
            assert err < 1e-12, err
        # the local matrix holds exactly this rank's blocks of the factor (2-D block-cyclic)
        for I in range(geo.pr, geo.nblk, geo.Pr):
            for J in range(geo.pc, geo.nblk, geo.Pc):
                if I >= J:
                    hi_, hj_ = min(geo.height(I), n - I * nb), min(geo.height(J), n - J * nb)
                    if hi_ <= 0 or hj_ <= 0:
                        continue
                    loc = run.A.a[(I // geo.Pr) * nb:(I // geo.Pr) * nb + hi_, (J // geo.Pc) * nb:(J // geo.Pc) * nb + hj_]
                    ref = Lref[I * nb:I * nb + hi_, J * nb:J * nb + hj_]
                    if I == J:
                        loc, ref = np.tril(loc), np.tril(ref)
                    assert np.max(np.abs(loc - ref)) <= 1e-12 * np.max(np.abs(Lref)), (I, J)
        alpha = run.ops.vec_to_host(run.alpha, n)
        assert np.max(np.abs(alpha - aref)) <= 1e-10 * np.max(np.abs(aref))
        llref = -0.5 * y @ aref - 0.5 * np.linalg.slogdet(Kref)[1] - n / 2.0 * np.log(2 * np.pi)
        assert abs(ll - llref) <= 1e-11 * abs(llref), (ll, llref)
        m_ref = orc.fit(spec, X, np.zeros(n), 0.05)
        ivref = abs(np.mean(orc.posterior(spec, m_ref, Z, compvar=1)[1]))
        assert abs(iv - ivref) <= 1e-10 * ivref, (iv, ivref)
    # non-positive-definite input: every rank reports the same pivot
    geo = run.geo
    ops.poison = False
    ops.kfill_local(None, X, run.A, 0.05, geo)
    if geo.owner_rank(0, 0) == comm.rank:
        run.A.a[5, 5] = -1.0
    info = dist.dist2_potrf(ops, comm, geo, run.A, run.G, L=run.L)
    assert info == 1, info
    if comm.rank == 0:
        print("DIST_OK cpu2d world=%d grid=%dx%d n=%d nb=%d maxerr=%.2e" % (comm.world, geo.Pr, geo.Pc, n, nb, err),
              flush=True)


class NumpyC5:
    """NumPy double of the device backend of dist_lml_grad / dist_mi_greedy (same four entry points + MiState): the slab sums
    straight from their definition on a dense inverse, the MI state from the definition of the down-date."""

    def __init__(self):
        from gpexp_amd import device as dev
        self.lml_grad_slab_bounds = dev.lml_grad_slab_bounds
        self.lml_grad_from_sums = dev.lml_grad_from_sums

    @staticmethod
    def _k(spec, A, B):
        d = spec.d
        cl, sig = spec.hyp[:d], spec.hyp[d]
        D = (A[:, None, :] - B[None, :, :]) / cl
        return sig * np.exp(-0.5 * np.sum(D * D, axis=2)), D

    def points(self, ctx, x):
        return np.asarray(x, dtype=float)

    def alloc_vector(self, ctx, n):
        return NumpyMat(np.zeros(n))

    def lml_grad_slab(self, ctx, spec, L, X, alpha, r0, r1):
        n, d = X.shape[0], spec.d
        K0, D = self._k(spec, X, X)
        P = np.linalg.inv(np.tril(L.a[:n, :n]) @ np.tril(L.a[:n, :n]).T)
        T = np.outer(alpha, alpha) - P
        out = np.zeros(d + 2)
        for a in range(r0, min(r1, n)):
            w = np.full(n - a, 2.0)
            w[0] = 1.0
            tk = w * T[a, a:] * K0[a, a:]
            for q in range(d):
                out[q] += np.sum(tk * D[a, a:, q] ** 2)
            out[d] += np.sum(tk)
            out[d + 1] += T[a, a]
        return out

    def lml_grad_rows_bounds(self, n, parts, nsub=1):
        from gpexp_amd import device as _real
        return _real.lml_grad_rows_bounds(n, parts, nsub)

    def lml_grad_rows(self, ctx, spec, L, X, alpha, r0, r1, nsub=1):
        """the rows [r0, r1) of L^-1: their share G = U_R^T U_R of K^-1 (gpx_lml_grad_rows); alpha alpha^T with the last slab"""
        n, d = X.shape[0], spec.d
        K0, D = self._k(spec, X, X)
        U = np.linalg.inv(np.tril(L.a[:n, :n]))
        R = U[r0:min(r1, n)]
        npad = L.a.shape[0]
        T = (np.outer(alpha, alpha) if r1 == npad else np.zeros((n, n))) - R.T @ R
        out = np.zeros(d + 2)
        for a in range(n):
            w = np.full(n - a, 2.0)
            w[0] = 1.0
            tk = w * T[a, a:] * K0[a, a:]
            for q in range(d):
                out[q] += np.sum(tk * D[a, a:, q] ** 2)
            out[d] += np.sum(tk)
            out[d + 1] += T[a, a]
        return out

    def MiState(self, ctx, spec, Cp, noise, nsel, start, lo, hi):
        be = self

        class St:
            def __init__(s_):
                s_.K0, _ = be._k(spec, Cp, Cp)
                s_.P = np.linalg.inv(s_.K0 + noise * np.eye(len(Cp)))
                s_.alive = np.ones(len(Cp), bool)
                s_.picks = [int(start)]
                s_.A = []

            def row(s_, cur, rowbuf):
                sidx = s_.picks[cur]
                if lo <= sidx < hi:
                    rowbuf.a[:len(Cp)] = s_.P[sidx]

            def score(s_, cur, rowbuf):
                sidx = s_.picks[cur]
                pr = rowbuf.a[:len(Cp)].copy()
                for i in range(lo, hi):
                    if i != sidx:
                        keep = s_.P[i, sidx]
                        s_.P[i] -= keep * pr / pr[sidx]
                        s_.P[i, sidx] = keep
                s_.alive[sidx] = False
                s_.A.append(sidx)
                A = s_.A
                KA = s_.K0[np.ix_(A, A)] + noise * np.eye(len(A))
                num = np.diag(s_.K0) - np.einsum("ia,ab,ib->i", s_.K0[:, A], np.linalg.inv(KA), s_.K0[:, A])
                ratio = np.full(len(Cp), -np.inf)
                for c in range(lo, hi):
                    if s_.alive[c]:
                        ratio[c] = num[c] / (1.0 / s_.P[c, c] - noise)
                if not np.isfinite(ratio).any():
                    return -np.inf, len(Cp)
                i = int(np.argmax(ratio))
                return float(ratio[i]), i

            def select(s_, slot, idx):
                s_.picks.append(int(idx))
        return St()


def run_cpu_c5(args):
    """BASELINE config C5's multi-GPU form over gloo with the NumPy doubles: distributed fit (NumPy device double), sharded
    log-marginal gradient (slab partition, rank-ordered sum) against the dense definition, sharded MI greedy (row owner,
    broadcast, first-max merge) against a single-process run of the same double."""
    from gpexp_amd import device as dev
    rng = np.random.default_rng(args.n)
    d, n, nb = 3, args.n, args.nb
    X = rng.uniform(-1, 1, (n, d))
    y = np.sin(2 * np.pi * X.sum(1) / d) + 0.2 * rng.standard_normal(n)
    hyp = np.array([0.4, 0.55, 0.7, 1.3])
    spec = dev.KernelSpec(dev.K_SE, d, hyp)
    spec_o = dict(kind="se", cl=list(hyp[:d]), signalSize=float(hyp[d]), d=d)
    comm = NumpyComm()
    grid = tuple(int(v) for v in args.grid.split("x")) if args.grid else dist.choose_grid(comm.world)
    ops = NumpyOps2D(spec_o)
    be = NumpyC5()
    Cand = rng.uniform(-1, 1, (47, d))
    run = dist.DistFitGrad2D(None, comm, spec, X, y, 0.05, nb=nb, ops=ops, grid=grid, cand=Cand, nsel=6, be=be)
    ll, grad, picks = run.step()
    from oracle import gpexp_oracle as orc
    K = orc.cov_matrix(spec_o, X, 0.05, row_loop=False)
    alpha = np.linalg.solve(K, y)
    P = np.linalg.inv(K)
    T = np.outer(alpha, alpha) - P
    K0, D = be._k(spec, X, X)
    ref = np.array([0.5 * np.sum(T * K0 * D[:, :, q] ** 2) / hyp[q] for q in range(d)] +
                   [0.5 * np.sum(T * K0) / hyp[d], 0.5 * np.trace(T)])
    assert np.max(np.abs(grad - ref)) <= 1e-9 * np.max(np.abs(ref)), (grad, ref)
    # slab boundaries: multiples of 128 covering the padded order, equal work to within one leaf row band
    b = dev.lml_grad_slab_bounds(n, comm.world)
    assert b[0] == 0 and b[-1] == dist.padded(n) and all(x % 128 == 0 for x in b) and b == sorted(b)
    # MI: sharded == one process holding every row
    class Solo:
        rank, world = 0, 1
        def bcast_grp(self, *a): pass
        def allgather(self, v): return np.atleast_2d(v)
    solo, _ = dist.dist_mi_greedy(None, Solo(), spec, Cand, 0.05, 6, be=be)
    assert list(picks) == list(solo), (picks, solo)
    assert len(set(picks)) == 6
    if comm.rank == 0:
        print("DIST_OK cpu-c5 world=%d grid=%dx%d n=%d grad0=%.6g picks=%s" % (comm.world, grid[0], grid[1], n, grad[0], list(picks)),
              flush=True)


def run_gpu_c5(args):
    """C5's multi-GPU form on the real HIP primitives (ranks share GPU 0, host-staged exchange): gradient = single-GPU
    gpx_lml_grad to 1e-10, MI picks identical to gpx_mi_greedy."""
    from gpexp_amd import device as dev
    ctx = dev.Context(int(os.environ.get("GPX_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    dev._ctx = ctx
    comm = dist.init_from_env(ctx)
    rng = np.random.default_rng(args.n)
    d, N = 5, args.n
    Xh = rng.uniform(-1, 1, (N, d))
    yh = np.sin(2 * np.pi * Xh.sum(1) / d) + 0.3 * rng.standard_normal(N)
    hyp = np.array([0.5, 0.53, 0.56, 0.59, 0.62, 1.0])
    spec = dev.KernelSpec(dev.K_SE, d, hyp)
    grid = tuple(int(v) for v in args.grid.split("x")) if args.grid else None
    Cand = rng.uniform(-1, 1, (args.m, d))
    run = dist.DistFitGrad2D(ctx, comm, spec, Xh, yh, 0.1, nb=args.nb, grid=grid, cand=Cand, nsel=7)
    ll, grad, picks = run.step()
    X = dev.points(ctx, Xh)
    K1 = dev.potrf(ctx, dev.kfill(ctx, spec, X, nugget=0.1))
    a1 = dev.potrs(ctx, K1, yh)
    g1 = dev.lml_grad(ctx, spec, K1, X, a1)
    assert np.max(np.abs(grad - g1)) <= 1e-10 * np.max(np.abs(g1)), (grad, g1)
    p1, r1 = dev.mi_greedy(ctx, spec, dev.points(ctx, Cand), 0.1, 7)
    assert list(picks) == list(p1), (picks, p1)
    comm.barrier()
    if comm.rank == 0:
        print("DIST_OK gpu-c5 world=%d n=%d comm=%s grad0=%.12g picks=%s" % (comm.world, N, type(comm).__name__, grad[0], list(picks)),
              flush=True)
    comm.close()
    ctx.close()


def _chaos_ops(ctx, seed, rank):
    """DeviceOps2D whose every primitive is preceded, with probability 1/3, by a kernel that holds the CURRENT stream back for
    1-40 ms (gpx_dbg_spin): results must not depend on how the streams of a rank -- or the ranks -- drift against each other.
    A dependency that is only satisfied by luck of timing (the k = 0 early update ran before the assembly once in a dozen
    fresh processes) fails under some seed instead of once in a blue moon."""
    from gpexp_amd._lib import check
    rng = np.random.default_rng(1000 * seed + rank)

    class ChaosOps2D(dist.DeviceOps2D):
        pass

    def wrap(name):
        f = getattr(dist.DeviceOps2D, name)

        def g(self, *a, **kw):
            if rng.random() < 0.4:
                check(self.ctx.lib.gpx_dbg_spin(self.ctx.h, int(rng.choice([1, 2, 5, 15, 40], p=[0.35, 0.3, 0.2, 0.1, 0.05]))))
            return f(self, *a, **kw)
        setattr(ChaosOps2D, name, g)
    for nm in ("kfill_local", "diag_factor", "panel_trsm", "update", "update_multi", "unpack_rows", "unpack_diag", "cross_fill", "ivar_group",
               "record"):
        wrap(nm)
    return ChaosOps2D(ctx)


def run_gpu2d_chaos(args):
    """First step of a FRESH runner under random per-stream delays, packed buffers and replicated factor pre-filled with NaN
    (a read of anything that has not arrived yet cannot hide behind zeros or last step's identical data), against the
    single-GPU path."""
    from gpexp_amd import device as dev
    ctx = dev.Context(int(os.environ.get("GPX_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    dev._ctx = ctx
    comm = dist.init_from_env(ctx)
    rng = np.random.default_rng(args.n)
    d = 4
    N, M = args.n, args.m
    Xh = rng.uniform(-1, 1, (N, d))
    yh = np.sin(2 * np.pi * Xh.sum(1) / d) + 0.3 * rng.standard_normal(N)
    Zh = rng.uniform(-1, 1, (M, d))
    spec = dev.KernelSpec(dev.K_MATERN52, d, [0.5, 1.0])
    grid = tuple(int(v) for v in args.grid.split("x")) if args.grid else None
    X = dev.points(ctx, Xh)
    K1 = dev.potrf(ctx, dev.kfill(ctx, spec, X, nugget=0.1))
    alpha1 = dev.potrs(ctx, K1, yh)
    ll1 = -0.5 * float(yh @ alpha1) - 0.5 * dev.logdet(ctx, K1) - N / 2.0 * np.log(2 * np.pi)
    iv1 = abs(dev.ivar(ctx, spec, K1, X, dev.points(ctx, Zh)))
    for streamed in (True, False):
        ops = _chaos_ops(ctx, args.chaos, comm.rank)
        runner = dist.DistFitIvar2D(ctx, comm, spec, Xh, yh, Zh, 0.1, nb=args.nb, streamed=streamed, grid=grid, ops=ops)
        from gpexp_amd._lib import check, dptr
        for buf in (runner.G[0], runner.G[1], runner.L):
            nan = np.full(buf.to_host().size, np.nan)
            check(ctx.lib.gpx_mat_write(ctx.h, buf.h, 0, nan.size, dptr(nan)))
        ll, iv = runner.step()
        assert np.isfinite(ll) and abs(ll - ll1) <= 1e-11 * abs(ll1), (streamed, ll, ll1)
        assert np.isfinite(iv) and abs(iv - iv1) <= 1e-11 * abs(iv1), (streamed, iv, iv1)
        del runner
    comm.barrier()
    if comm.rank == 0:
        print("DIST_OK gpu2d-chaos world=%d seed=%d n=%d nb=%d ll=%.12g ivar=%.12g" % (comm.world, args.chaos, N, args.nb, ll, iv),
              flush=True)
    comm.close()
    ctx.close()


def run_gpu2d(args):
    """2-D path on the real HIP primitives: ranks share GPU 0 and exchange through the host-staged gloo communicator."""
    from gpexp_amd import device as dev
    ctx = dev.Context(int(os.environ.get("GPX_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    dev._ctx = ctx
    comm = dist.init_from_env(ctx)
    rng = np.random.default_rng(args.n)
    d = 4
    N, M = args.n, args.m
    Xh = rng.uniform(-1, 1, (N, d))
    yh = np.sin(2 * np.pi * Xh.sum(1) / d) + 0.3 * rng.standard_normal(N)
    Zh = rng.uniform(-1, 1, (M, d))
    spec = dev.KernelSpec(dev.K_MATERN52, d, [0.5, 1.0])
    grid = tuple(int(v) for v in args.grid.split("x")) if args.grid else None
    runner = dist.DistFitIvar2D(ctx, comm, spec, Xh, yh, Zh, 0.1, nb=args.nb, grid=grid)
    ll, iv = runner.step()
    ll2, iv2 = runner.step()  # second step re-assembles in place
    assert ll == ll2 and iv == iv2, (ll, ll2, iv, iv2)
    other = dist.DistFitIvar2D(ctx, comm, spec, Xh, yh, Zh, 0.1, nb=args.nb, streamed=not runner.streamed, grid=grid)
    ll3, iv3 = other.step()
    # the streamed schedule keeps a window of the factor, the other one the replica
    assert bool(runner.window) == runner.streamed and bool(other.window) == other.streamed
    Ld = (other if runner.window else runner).L.to_host(tri=1)
    del other
    X = dev.points(ctx, Xh)
    K1 = dev.kfill(ctx, spec, X, nugget=0.1)
    dev.potrf(ctx, K1)
    L1 = K1.to_host(tri=1)
    errL = float(np.max(np.abs(Ld - L1)) / np.max(np.abs(L1)))
    alpha1 = dev.potrs(ctx, K1, yh)
    ll1 = -0.5 * float(yh @ alpha1) - 0.5 * dev.logdet(ctx, K1) - N / 2.0 * np.log(2 * np.pi)
    iv1 = abs(dev.ivar(ctx, spec, K1, X, dev.points(ctx, Zh)))
    # both evaluation schedules (streamed underneath the factorisation / after it) against the single-GPU path
    assert abs(iv3 - iv1) <= 1e-11 * abs(iv1) and abs(ll3 - ll) <= 1e-12 * abs(ll), (runner.streamed, iv, iv3, iv1, ll, ll3)
    assert errL < 1e-12, errL
    alpha = runner.ops.vec_to_host(runner.alpha, N)
    assert np.max(np.abs(alpha - alpha1)) <= 1e-10 * np.max(np.abs(alpha1))
    assert abs(ll - ll1) <= 1e-11 * abs(ll1), (ll, ll1)      # every rank has the log-likelihood (all-reduced pieces)
    assert abs(iv - iv1) <= 1e-11 * abs(iv1), (iv, iv1)
    errs = comm.allgather(np.array([errL]))[:, 0]
    comm.barrier()
    if comm.rank == 0:
        print("DIST_OK gpu2d world=%d grid=%dx%d n=%d nb=%d comm=%s errL=%.2e ll=%.12g ivar=%.12g" %
              (comm.world, runner.geo.Pr, runner.geo.Pc, N, args.nb, type(comm).__name__, float(errs.max()), ll, iv),
              flush=True)
    comm.close()
    ctx.close()


def run_gpu(args):
    from gpexp_amd import device as dev
    ctx = dev.Context(int(os.environ.get("GPX_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    dev._ctx = ctx
    comm = dist.init_from_env(ctx)
    rng = np.random.default_rng(args.n)
    d = 4
    N, M = args.n, args.m
    Xh = rng.uniform(-1, 1, (N, d))
    yh = np.sin(2 * np.pi * Xh.sum(1) / d) + 0.3 * rng.standard_normal(N)
    Zh = rng.uniform(-1, 1, (M, d))
    spec = dev.KernelSpec(dev.K_MATERN52, d, [0.5, 1.0])
    runner = dist.DistFitIvar(ctx, comm, spec, Xh, yh, Zh, 0.1, nb=args.nb)
    ll, iv = runner.step()
    ll2, iv2 = runner.step()  # second step re-assembles in place
    assert ll == ll2 and iv == iv2
    # the other evaluation schedule (streamed <-> after the factorisation) must agree
    other = dist.DistFitIvar(ctx, comm, spec, Xh, yh, Zh, 0.1, nb=args.nb, streamed=not runner.streamed)
    ll3, iv3 = other.step()
    assert abs(iv3 - iv) <= 1e-11 * abs(iv), (runner.streamed, iv, iv3)
    if comm.rank == 0:
        assert abs(ll3 - ll) <= 1e-12 * abs(ll), (ll, ll3)
    del other
    Ld = runner.K.to_host(tri=1)
    # single-GPU path on the same inputs (every rank checks its own copy of L)
    X = dev.points(ctx, Xh)
    K1 = dev.kfill(ctx, spec, X, nugget=0.1)
    dev.potrf(ctx, K1)
    L1 = K1.to_host(tri=1)
    errL = float(np.max(np.abs(Ld - L1)) / np.max(np.abs(L1)))
    alpha = dev.potrs(ctx, K1, yh)
    ll1 = -0.5 * float(yh @ alpha) - 0.5 * dev.logdet(ctx, K1) - N / 2.0 * np.log(2 * np.pi)
    iv1 = abs(dev.ivar(ctx, spec, K1, X, dev.points(ctx, Zh)))
    assert errL < 1e-12, errL
    if comm.rank == 0:  # the log-likelihood is produced on rank 0 only (its sweeps hide under rank 0's evaluation slice)
        assert abs(ll - ll1) <= 1e-11 * abs(ll1), (ll, ll1)
    assert abs(iv - iv1) <= 1e-11 * abs(iv1), (iv, iv1)
    # sharded design search: candidates split over the ranks, first-minimum merge == single-GPU selection
    Ch = rng.uniform(-1, 1, (517, d))
    Zd = dev.points(ctx, Zh)
    gidx, gcost = dist.dist_greedy_ivar_step(ctx, comm, spec, runner.K, runner.X, Ch, Zd, 0.1)
    sbest, scosts = dev.greedy_ivar_step(ctx, spec, K1, X, dev.points(ctx, Ch), Zd, 0.1)
    assert gidx == sbest and abs(gcost - scosts[sbest]) <= 1e-12 * abs(gcost), (gidx, sbest, gcost, scosts[sbest])
    # exact tie between two candidates living on different ranks: the lower global index must win
    Ct = Ch.copy()
    Ct[-1] = Ct[sbest]
    if sbest != len(Ct) - 1:
        tidx, _ = dist.dist_greedy_ivar_step(ctx, comm, spec, runner.K, runner.X, Ct, Zd, 0.1)
        assert tidx == sbest, (tidx, sbest)
    errs = comm.allgather(np.array([errL]))[:, 0]
    comm.barrier()
    if comm.rank == 0:
        print("DIST_OK gpu world=%d n=%d nb=%d comm=%s errL=%.2e ll=%.12g ivar=%.12g" %
              (comm.world, N, args.nb, type(comm).__name__, float(errs.max()), ll, iv), flush=True)
    comm.close()
    ctx.close()


def run_cpu_api(args):
    import api_worker
    api_worker.run_api(args, gpu=False)


def run_gpu_api(args):
    import api_worker
    api_worker.run_api(args, gpu=True)


def run_gpu_api_c4lite(args):
    import api_worker
    api_worker.run_api_c4lite(args)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="cpu")
    ap.add_argument("--npts", dest="n", type=int, default=700)
    ap.add_argument("--mpts", dest="m", type=int, default=333)
    ap.add_argument("--blk", dest="nb", type=int, default=256)
    ap.add_argument("--grid", default="")
    ap.add_argument("--chaos", type=int, default=0)
    a = ap.parse_args()
    print("WORKER_UP rank=%s mode=%s" % (os.environ.get("RANK", "0"), a.mode), flush=True)
    {"cpu": run_cpu, "gpu": run_gpu, "cpu2d": run_cpu2d, "gpu2d": run_gpu2d, "gpu2d-chaos": run_gpu2d_chaos,
     "cpu-c5": run_cpu_c5, "gpu-c5": run_gpu_c5, "cpu-api": run_cpu_api, "gpu-api": run_gpu_api,
     "gpu-api-c4lite": run_gpu_api_c4lite}[a.mode](a)
    if "torch" in sys.modules:  # orderly gloo teardown: a rank that exits while its peers still hold sub-group
        import torch.distributed as td   # connections aborts in a gloo thread (the RCCL path never imports torch)
        if td.is_initialized():
            td.barrier()
            td.destroy_process_group()
