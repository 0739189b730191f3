"""CPU-only tests (-m "not gpu"): the C-ABI library builds, loads and exports every symbol include/gpx.h declares;
the product path has no CPU fallback and never touches the oracle; host-side argument logic."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session")
def built_lib():
    so = os.path.join(ROOT, "gpexp_amd", "libgpx_hip.so")
    csrc = os.path.join(ROOT, "gpexp_amd", "csrc")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h"))] + \
           [os.path.join(ROOT, "include", h) for h in ("gpx.h", "gpx_dist.h", "gpx_debug.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        if not os.path.exists("/opt/rocm/bin/hipcc"):
            pytest.skip("hipcc not available and library not built")
        subprocess.run(["make", "-C", csrc, "-j4"], check=True, stdout=subprocess.DEVNULL)
    return so


def header_symbols(which=("gpx.h", "gpx_dist.h", "gpx_debug.h")):
    """Functions declared by include/gpx.h (the drop-in ABI), include/gpx_dist.h (multi-GPU / scheduler primitives that only
    gpexp_amd/dist.py drives) and include/gpx_debug.h (test hooks)."""
    out = set()
    for name in which:
        txt = open(os.path.join(ROOT, "include", name)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        out |= set(re.findall(r"\b(gpx_[a-z0-9_]+)\s*\(", txt))
    return sorted(out)


def test_header_symbols_all_exported(built_lib):
    import ctypes
    lib = ctypes.CDLL(built_lib)
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), "include/gpx.h declares %s but libgpx_hip.so does not export it" % s


def test_binding_matches_header(built_lib):
    from gpexp_amd import _lib
    assert sorted(_lib.exported_symbols()) == header_symbols()
    lib = _lib.load()
    assert lib.gpx_abi_version() == 2


def test_public_header_has_no_debug_hooks():
    assert not [f for f in header_symbols(("gpx.h",)) if f.startswith("gpx_dbg_")]
    assert all(f.startswith("gpx_dbg_") for f in header_symbols(("gpx_debug.h",)))


def test_drop_in_header_is_free_of_scheduler_internals():
    """include/gpx.h is the drop-in ABI -- one entry point per reference call site -- and nothing in it is a
    primitive that only gpexp_amd/dist.py can drive; those (streams / events, communicators, panel primitives, recorded
    programs, sharded state machines) are declared in include/gpx_dist.h, which includes gpx.h."""
    pub = header_symbols(("gpx.h",))
    internals = ("gpx_dist", "gpx_comm_", "gpx_program_", "gpx_graph_", "gpx_event_", "gpx_stream_", "gpx_mi_begin", "gpx_mi_row",
                 "gpx_mi_score", "gpx_mi_select", "gpx_mi_end", "gpx_givar_", "gpx_lml_grad_slab", "gpx_vec_op", "gpx_mat_read",
                 "gpx_mat_write")
    assert not [f for f in pub if f.startswith(internals)], [f for f in pub if f.startswith(internals)]
    dist = header_symbols(("gpx_dist.h",))
    assert len(dist) >= 50 and not set(dist) & set(pub)
    assert '#include "gpx.h"' in open(os.path.join(ROOT, "include", "gpx_dist.h")).read()
    # everything the class API of ONE process calls is in gpx.h (gpexp_amd/device.py minus the sharded state machines)
    for f in ("gpx_kfill", "gpx_potrf", "gpx_potrs", "gpx_logdet", "gpx_posterior", "gpx_ivar", "gpx_greedy_var",
              "gpx_greedy_ivar_step", "gpx_greedy_ivar", "gpx_mi_greedy", "gpx_lml_grad", "gpx_lml_grad_linv", "gpx_mat_clone"):
        assert f in pub, f


def test_packed_row_stride_constant_matches_library(built_lib):
    from gpexp_amd import _lib, dist
    lib = _lib.load()
    for nb in (128, 256, 512, 1024):
        assert lib.gpx_dist2_row_stride(nb) == nb + dist.G_SKEW


def test_header_cites_reference_lines():
    txt = open(os.path.join(ROOT, "include", "gpx.h")).read()
    for cite in ("gp_kernel_utilities.py:34-68", "gp.py:181", "gp.py:434", "experimentalDesign.py:787-845",
                 "kernels.py:49-65"):
        assert cite in txt


def test_fails_loudly_without_gpu(built_lib):
    import ctypes
    lib = ctypes.CDLL(built_lib)
    n = ctypes.c_int(0)
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        have_gpu = hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        have_gpu = False
    if have_gpu:
        pytest.skip("a GPU is visible")
    from gpexp_amd import device
    with pytest.raises(RuntimeError, match="MI355X"):
        device.Context(0)
    from gpexp_amd.kernels import KernelSquaredExponential
    from gpexp_amd.gp import GP
    g = GP(KernelSquaredExponential([0.3], 1.0, 1), 0.1)
    with pytest.raises(RuntimeError):
        g.train(np.zeros((3, 1)), np.zeros(3))  # no silent NumPy fallback


def test_product_never_imports_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|gpexp_oracle", re.M)
    for base in ("gpexp_amd", "gpExp"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp")):
                    src = open(os.path.join(dirpath, f)).read()
                    assert not pat.search(src), "%s/%s references the oracle" % (dirpath, f)


def test_shape_assertions_before_any_device_work(built_lib):
    from gpexp_amd.kernels import KernelSquaredExponential, KernelMehler1D, KernelIsoMatern
    from gpexp_amd.gp import GP
    k = KernelSquaredExponential([0.3], 1.0, 2)
    with pytest.raises(AssertionError):
        k.evaluate(np.zeros(4), np.zeros((1, 2)))
    with pytest.raises(AssertionError):
        k.evaluate(np.zeros((4, 3)), np.zeros((1, 3)))
    with pytest.raises(AssertionError):
        k.evaluate(np.zeros((4, 2)), np.zeros((3, 2)))
    with pytest.raises(AssertionError):
        KernelMehler1D(0.5, 2)
    with pytest.raises(AssertionError):
        GP(k, 0.1).train(np.zeros((3, 2)), np.zeros((3, 1)))
    with pytest.raises(AssertionError):
        GP(k, 0.1).evaluateVariance(np.zeros((3, 2)))  # no training points yet
    with pytest.raises(AttributeError):
        KernelIsoMatern(0.5, 1.0, 2).derivativeWrtHypParams(np.zeros((2, 2)), np.zeros((2, 2)))
    g = GP(k, 0.1, FITC=0.5)          # f4: accepted; the attribute is stored as in the reference (gp.py:69-70)
    assert g.FITC == 0.5 and g.fitcnodes is None
    with pytest.raises(NotImplementedError):
        GP(k, 0.1).generateSamples(np.zeros((2, 2)))


def test_kernel_hyperparameter_dicts(built_lib):
    from gpexp_amd.kernels import KernelSquaredExponential, KernelIsoMatern, KernelMehlerND
    from gpexp_amd import device as dev
    k = KernelSquaredExponential([0.4, 0.9], 2.0, 2)
    assert k.hyperParam == {"cl0": 0.4, "cl1": 0.9, "signalSize": 2.0}
    sp = k._spec()
    assert (sp.kind, sp.d, list(sp.hyp)) == (dev.K_SE, 2, [0.4, 0.9, 2.0])
    iso = KernelSquaredExponential([0.3], 1.0, 3)  # length-1 correlation length is tiled (kernels.py:106-107)
    assert list(iso._spec().hyp) == [0.3, 0.3, 0.3, 1.0]
    m = KernelIsoMatern(0.7, 1.5, 2)
    assert m.hyperParam == {"rho": 0.7, "signalSize": 1.5} and m._spec().kind == dev.K_MATERN32
    assert KernelIsoMatern(0.7, 1.5, 2, nu=2.5)._spec().kind == dev.K_MATERN52
    me = KernelMehlerND([0.5, 0.3], 2)
    assert me.hyperParam == {0: 0.5, 1: 0.3} and me._spec().kind == dev.K_MEHLER


def test_nugget_type_rules(built_lib):
    from gpexp_amd import device as dev
    assert dev._nugget_args(0.0, 5) == (None, 0)
    a, n = dev._nugget_args(0.1, 5)
    assert n == 1 and a[0] == 0.1
    a, n = dev._nugget_args(np.arange(5.0), 5)
    assert n == 5
    with pytest.raises(TypeError):
        dev._nugget_args(1, 5)
    with pytest.raises(AssertionError):
        dev._nugget_args(np.arange(4.0), 5)


def test_gpexp_shim_exports_reference_names(built_lib):
    import gpExp.experimentalDesign as ed
    import gpExp.kernels as kn
    import gpExp.gp as gpm
    import gpExp.approximation as ap
    import gpExp.gp_kernel_utilities as ku
    for nm in ("costFunctionBase", "costFunctionGP_IVAR", "costFunctionGP_MI", "ExperimentalDesign",
               "ExperimentalDesignDerivative", "performGreedyVarExperimentalDesign",
               "performGreedyMIExperimentalDesign"):
        assert hasattr(ed, nm)
    for nm in ("Kernel", "KernelSquaredExponential", "KernelIsoMatern", "KernelMehlerND", "KernelMehler1D"):
        assert hasattr(kn, nm)
    assert hasattr(gpm, "GP") and hasattr(ap, "Space") and hasattr(ku, "calculateCovarianceMatrix")
    ns = {}
    exec("from gpExp.experimentalDesign import *", ns)  # demo.py:27
    assert "costFunctionGP_IVAR" in ns and "ExperimentalDesignDerivative" in ns


def test_column_reduction_scratch_bound_covers_every_sub_block():
    """One scratch buffer serves all the sub-blocks a triangular sweep reduces, so the bound callers allocate must dominate
    the need of every smaller launch.  (Sizing it with the exact need of the LARGEST launch did not: a 16640 x 16384 block
    needs 524288 partial sums, the 33024 x 33024 launch only 495360 -- the sweeps of a 33024-point factor wrote 230 KB past
    their scratch.)  Host logic only: the library's own planner through its test hook."""
    import ctypes as C
    from gpexp_amd import _lib
    lib = _lib.load()
    nch, bnd = C.c_int64(), C.c_int64()

    def plan(r, p):
        assert lib.gpx_dbg_colreduce_plan(r, p, C.byref(nch), C.byref(bnd)) == 0
        return nch.value, bnd.value

    rng = np.random.default_rng(11)
    for R, P in [(33024, 33024), (65664, 65664), (9216, 9216), (2176, 2176), (128, 128), (70000, 300)]:
        _, bound = plan(R, P)
        for _ in range(3000):
            r, p = int(rng.integers(1, R + 1)), int(rng.integers(1, P + 1))
            n, own = plan(r, p)
            assert n * p <= own <= bound and n <= 2048
    # the shapes that overflowed
    assert plan(16640, 16384)[0] * 16384 <= plan(33024, 33024)[1]
    assert plan(5120, 4096)[0] * 4096 <= plan(9216, 9216)[1]


def test_all_link_panel_broadcast_schedule_delivers_every_piece():
    """gpx_comm_panel_bcast (scatter + all-gather over grouped ncclSend / ncclRecv) has only ever run with one rank on
    hardware.  Its schedule is host logic: replay it for every rank of communicators of 2..8 ranks with the semantics RCCL
    gives a group -- the i-th send from a to b pairs with the i-th receive on b from a, lengths must match, all transfers of a
    group read the state from before the group -- and check that every rank ends up with every piece (root's data), for piece
    sets like the 2-D loop's (Pr pieces with different roots, ragged and empty ones, both sides of the direct-send threshold)."""
    import ctypes as C
    from gpexp_amd import _lib
    lib = _lib.load()

    def plan(W, me, small, counts, roots):
        cnt = (C.c_int64 * len(counts))(*counts)
        rts = (C.c_int * len(roots))(*roots)
        n = C.c_int64()
        assert lib.gpx_dbg_panel_bcast_plan(W, me, small, len(counts), cnt, rts, None, 0, C.byref(n)) == 0
        ops = (C.c_int64 * (6 * max(n.value, 1)))()
        assert lib.gpx_dbg_panel_bcast_plan(W, me, small, len(counts), cnt, rts, ops, n.value, C.byref(n)) == 0
        return [tuple(ops[6 * i:6 * i + 6]) for i in range(n.value)]

    rng = np.random.default_rng(5)
    cases = 0
    for W in range(2, 9):
        for trial in range(12):
            npieces = int(rng.integers(1, 5))
            counts = [int(rng.choice([0, 2, 130, 4096, 70000, 70001, 262144 + 2 * int(rng.integers(0, 50))])) for _ in range(npieces)]
            roots = [int(rng.integers(0, W)) for _ in range(npieces)]
            small = int(rng.choice([1, 65536]))
            # every rank's buffer: piece i holds (rank, i, index) tagged data only on its root, NaN elsewhere
            data = [[np.full(c, np.nan) for c in counts] for _ in range(W)]
            for i, (c, r) in enumerate(zip(counts, roots)):
                data[r][i] = 1e6 * (i + 1) + np.arange(c, dtype=float)
            plans = [plan(W, me, small, counts, roots) for me in range(W)]
            for phase in (1, 2):
                before = [[a.copy() for a in d] for d in data]
                sends = {}
                recvs = {}
                for me in range(W):
                    for (ph, is_send, piece, off, ln, peer) in plans[me]:
                        if ph != phase:
                            continue
                        assert 0 <= peer < W and peer != me and ln > 0 and off >= 0 and off + ln <= counts[piece]
                        (sends if is_send else recvs).setdefault((me, peer) if is_send else (peer, me), []).append((piece, off, ln))
                assert set(sends) == set(recvs), (W, counts, roots, phase)
                for key in sends:                      # key = (source, destination)
                    assert len(sends[key]) == len(recvs[key])
                    for (sp, so, sl), (rp, ro, rl) in zip(sends[key], recvs[key]):
                        assert sl == rl                # RCCL pairs them in order; a length mismatch corrupts or hangs
                        data[key[1]][rp][ro:ro + rl] = before[key[0]][sp][so:so + sl]
            for me in range(W):
                for i, c in enumerate(counts):
                    assert np.array_equal(data[me][i], 1e6 * (i + 1) + np.arange(c, dtype=float)), (W, me, i, counts, roots, small)
            cases += 1
    assert cases == 7 * 12


def test_file_rendezvous_three_ranks(tmp_path):
    """The RCCL path exchanges its ncclUniqueId through FileRendezvous (no framework in those processes); only one rank ever
    used it on hardware.  Three processes with the launcher's environment, rank 0 starting LAST: all end with rank 0's 128
    bytes, and the file is gone after rank 0's cleanup."""
    import hashlib
    import subprocess
    import sys
    import time
    code = ("import os, sys, time, hashlib\n"
            "sys.path.insert(0, %r)\n"
            "from gpexp_amd.dist import FileRendezvous\n"
            "r = FileRendezvous()\n"
            "if r.rank == 0: time.sleep(0.5)\n"
            "blob = bytes([7 * i %% 256 for i in range(128)]) if r.rank == 0 else b'x' * 128\n"
            "got = r.exchange(blob)\n"
            "print(r.rank, hashlib.sha1(got).hexdigest(), len(got), flush=True)\n"
            "time.sleep(0.3 if r.rank == 0 else 0.0)\n"
            "r.cleanup()\n" % ROOT)
    env = dict(os.environ, WORLD_SIZE="3", MASTER_PORT="45123", TORCHELASTIC_RUN_ID="unit", GPX_RDV_DIR=str(tmp_path))
    procs = [subprocess.Popen([sys.executable, "-c", code], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, text=True)
             for r in (2, 1, 0)]
    outs = [p.communicate(timeout=120)[0].split() for p in procs]
    assert all(p.returncode == 0 for p in procs)
    want = hashlib.sha1(bytes([7 * i % 256 for i in range(128)])).hexdigest()
    assert sorted(o[0] for o in outs) == ["0", "1", "2"] and all(o[1] == want and o[2] == "128" for o in outs)
    time.sleep(0.1)
    assert list(tmp_path.iterdir()) == []


def test_declared_rccl_abi_matches_the_installed_header():
    """dist.hip binds RCCL through dlopen with hand-declared function types; most of those entry points have never been
    called with more than one rank.  tests/abi/rccl_abi_check.cpp pins every declared type, constant and size against
    <rccl/rccl.h> with static_asserts (compile-only, no GPU)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc) or not os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        pytest.skip("no hipcc / rccl.h on this machine")
    r = subprocess.run([hipcc, "-std=c++17", "-fsyntax-only", "-x", "hip", "--offload-arch=gfx950", "-I/opt/rocm/include",
                        os.path.join(ROOT, "tests", "abi", "rccl_abi_check.cpp")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]


def test_lml_grad_falls_back_when_the_big_scratch_does_not_fit(monkeypatch):
    """dev.lml_grad picks the L^-1 form from a static size test, not from free memory -- an out-of-memory error of that
    form (2 N^2 + N^2 / 4 doubles of scratch) must fall through to the rows form (one N x N accumulator) and then to the slab loop
    (no N x N buffer), while any OTHER error propagates.  Host logic only: the device entry points are replaced."""
    import numpy as np
    from gpexp_amd import device as dev
    from gpexp_amd._lib import GpxError

    class Ctx:
        trims = 0

        def trim(self):
            Ctx.trims += 1

    class Pts:
        shape = (60000, 3)

    spec = dev.KernelSpec(dev.K_SE, 3, [0.5, 0.5, 0.5, 1.0])
    calls = []

    def oom(*a, **k):
        calls.append("oom")
        raise GpxError("libgpx_hip: hipMalloc(68719476736 bytes) failed: out of memory")

    monkeypatch.setattr(dev, "lml_grad_linv_fits", lambda ctx, n: True)
    monkeypatch.setattr(dev, "lml_grad_linv", oom)
    monkeypatch.setattr(dev, "lml_grad_rows", oom)
    monkeypatch.setattr(dev, "lml_grad_slab", lambda ctx, spec, L, X, a, r0, r1: (calls.append((r0, r1)), np.ones(spec.d + 2))[1])
    monkeypatch.setenv("GPX_LML_GRAD_FORM", "linv")
    g = dev.lml_grad(Ctx(), spec, None, Pts(), np.zeros(60000))
    assert calls[:2] == ["oom", "oom"] and len(calls) > 2 and Ctx.trims == 2 and g.shape == (5,)
    slabs = [c for c in calls if c != "oom"]
    assert slabs[0][0] == 0 and slabs[-1][1] == 60032 and all(a[1] == b[0] for a, b in zip(slabs, slabs[1:]))

    def broken(*a, **k):
        raise GpxError("libgpx_hip: bad argument: X does not match the factor")

    monkeypatch.setattr(dev, "lml_grad_linv", broken)
    with pytest.raises(GpxError, match="does not match"):
        dev.lml_grad(Ctx(), spec, None, Pts(), np.zeros(60000))
