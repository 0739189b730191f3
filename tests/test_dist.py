"""Multi-rank tests.  CPU (-m "not gpu"): pure index logic + the real panel loops over gloo with a NumPy device double:
the 2-D block-cyclic path (north_star) at world 2, 4, 6 and 8 on 1x2 / 2x2 / 2x3 / 2x4 grids with ragged N, and the 1-D
block-column path at world 2 and 3.  GPU (-m gpu): real kernels, 2 to 6 ranks sharing the GPU through the host-staged
communicator, and the RCCL communicator (ncclCommSplit, grouped send/recv) with world_size 1 -- RCCL with more than one
rank needs more than one GPU: the `*_real_rccl` cases at the end switch themselves on when the box shows that many devices
(one rank per device, GPX_COMM=rccl, same comparisons as the host-staged ones) and are collected-and-skipped otherwise."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(world, extra, env_extra=None, timeout=600):
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "OMP_NUM_THREADS": "2", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    env.update(env_extra or {})
    for attempt in range(2):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
               os.path.join(ROOT, "tests", "dist_worker.py")] + extra
        r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
        # the port is picked, released and only then bound by the launcher: a rendezvous that lost that race is retried
        # once on a fresh port -- but ONLY when no worker got as far as its first line (every worker prints WORKER_UP
        # right after the process group is up), so a rank dying mid-run is never retried into a pass
        started = "WORKER_UP" in r.stdout or "WORKER_UP" in r.stderr
        if r.returncode == 0 or started or not any(m in r.stderr for m in ("EADDRINUSE", "Address already in use",
                                                                             "DistNetworkError", "Connection refused")):
            break
        print("launcher retry after rendezvous failure:\n" + r.stderr[-1500:], file=sys.stderr)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "DIST_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    # GPX_ALLOC_GUARD=1 in the environment reaches the workers too: their violations only show on stderr
    assert "ALLOCATION GUARD VIOLATED" not in r.stderr, r.stderr[-3000:]
    return r.stdout


def test_default_block_size_and_aggregation():
    """default_nb / default_agg: decided from (N, world, evaluation schedule) alone -- the same on every rank -- and the product of
    the two (the k range of a trailing update) stays 2048 where a rank is throughput-bound and 1024 where the chain across ranks
    bounds the grid."""
    from gpexp_amd import dist
    assert [dist.default_nb(n, 8) for n in (1000, 2048, 8192, 16384, 32768)] == [128, 256, 512, 1024, 1024]
    assert dist.default_nb(32768, 4) == 512 and dist.default_nb(32768, 8, streamed=True) == 512 and dist.default_nb(131072, 8, True) == 512
    for world, streamed in ((1, False), (2, False), (4, False), (8, False), (8, True)):
        nb = dist.default_nb(32768, world, streamed)
        assert dist.default_agg(streamed, world, nb) * nb == (2048 if (streamed or world < 4) else 1024)


def test_index_logic():
    from gpexp_amd import dist
    assert dist.padded(1) == 128 and dist.padded(128) == 128 and dist.padded(129) == 256
    assert dist.num_blocks(700, 256) == 3 and dist.num_blocks(32768, 512) == 64
    for world in (1, 2, 3, 8):
        blocks = sorted(sum((dist.owned_blocks(5000, 512, r, world) for r in range(world)), []))
        assert blocks == list(range(dist.num_blocks(5000, 512)))  # every block column has exactly one owner
        sl = [dist.eval_slice(1001, r, world) for r in range(world)]
        assert sl[0][0] == 0 and sl[-1][1] == 1001
        assert all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
        assert max(b - a for a, b in sl) - min(b - a for a, b in sl) <= 1
    assert dist.eval_slice(2, 3, 4) == (2, 2)  # more ranks than points: empty slice
    assert dist.panel_elems(700, 256) == 768 * 256 + 2 * 128 * 128
    assert dist.ordered_sum([0.1, 0.2, 0.3]) == (0.1 + 0.2) + 0.3
    # first-minimum rule across ranks (np.argmin semantics: ties -> lowest global index)
    assert dist.merge_argmin([0.5, 0.2, 0.2], [7, 40, 12]) == (0.2, 12)
    assert dist.merge_argmin([np.inf, 0.3], [99, 5]) == (0.3, 5)  # a rank with an empty slice reports +inf


@pytest.mark.parametrize("world,n,nb", [(2, 700, 256), (3, 1000, 128), (2, 300, 512)])
def test_panel_loop_gloo_cpu(world, n, nb):
    out = launch(world, ["--mode", "cpu", "--npts", str(n), "--blk", str(nb)])
    assert "world=%d" % world in out


@pytest.mark.parametrize("world,n,nb,grid", [(2, 700, 128, ""), (4, 1000, 256, ""), (6, 900, 128, ""), (8, 1300, 128, ""),
                                             (8, 333, 256, ""), (4, 1100, 256, "4x1"), (4, 1700, 256, ""), (8, 1300, 128, "4x2")])
def test_panel_loop_2d_gloo_cpu(world, n, nb, grid):
    """North-star layout: Pr x Pc block-cyclic ownership, column-communicator diagonal broadcast, all-rank panel pieces,
    look-ahead order, replicated + distributed factor, distributed substitution (reduce / bcast on sub-groups),
    all-reduced log-determinant, streamed evaluation hook -- on gloo with the NumPy device double."""
    out = launch(world, ["--mode", "cpu2d", "--npts", str(n), "--blk", str(nb), "--grid", grid], timeout=900)
    assert "world=%d" % world in out
    if not grid:
        from gpexp_amd import dist
        assert "grid=%dx%d" % dist.choose_grid(world) in out


def test_panel_loop_2d_fewer_points_than_ranks_gloo_cpu():
    """Three evaluation points on four ranks: one rank streams nothing (no window, no hook), the others do; the forward
    substitution then does NOT ride along (all ranks agree from (M, world)) and alpha comes from the full distributed sweeps."""
    out = launch(4, ["--mode", "cpu2d", "--npts", "900", "--mpts", "3", "--blk", "128"], timeout=900)
    assert "world=4" in out


@pytest.mark.parametrize("world,n,nb,grid,agg,bulk", [(4, 1700, 128, "", 4, "chunks"), (4, 1500, 128, "", 2, "chunks"),
                                                     (2, 1700, 128, "", 3, "eval"), (8, 1300, 128, "", 8, "eval"),
                                                     (4, 1500, 128, "", 1, "main"), (6, 1300, 128, "", 4, "bulk"),
                                                     (2, 1700, 128, "", 5, "chunks"), (1, 1500, 128, "", 4, "eval")])
def test_panel_loop_2d_schedules_gloo_cpu(world, n, nb, grid, agg, bulk):
    """The aggregated schedule of dist2_potrf in every mode: group sizes 1..8 (panels per trailing update; ragged last
    groups), the bulk update cut into chunks behind the near updates on MAIN / whole on a second stream, buffer-ring reuse
    (2 x agg packed buffers against 12-17 panels, so buffers are rewritten; the NumPy double starts every buffer NaN-filled and
    asserts on reads of anything unwritten, and a stale read of an older panel's bytes shows as a wrong factor).  These runs
    also take alpha / log-det from the block-cyclic factor by DISTRIBUTED substitution in every variant (GPX_DIST_SOLVE=dist;
    by default a rank that holds a replica of the factor uses that)."""
    out = launch(world, ["--mode", "cpu2d", "--npts", str(n), "--blk", str(nb), "--grid", grid],
                 {"GPX_DIST_AGG": str(agg), "GPX_DIST_BULK": bulk, "GPX_DIST_SOLVE": "dist"}, timeout=900)
    assert "world=%d" % world in out


@pytest.mark.parametrize("world,n,nb,env", [
    (4, 1500, 128, {"GPX_DIST2_STAGED_DIAG": "0"}),                              # diagonal block factored out of the local matrix (round 3)
    (4, 1500, 128, {"GPX_DIST_GATE_BULK": "0"}), (4, 1300, 128, {"GPX_DIST_GATE_BULK": "1"}),
    (4, 1500, 128, {"GPX_DIST2_HOIST_INV": "0", "GPX_DIST2_LATE_COPYBACK": "0"}),
    (2, 1300, 128, {"GPX_DIST_EARLY_BUF": "0", "GPX_DIST_AGG": "1"})])
def test_panel_loop_2d_round4_knobs_gloo_cpu(world, n, nb, env):
    """Round 4 moved work off the chain across ranks step by step, each step behind a switch (staged diagonal block, deferred bulk
    update, hoisted inverse + late copy-back, the early block row's own buffer): the older forms stay valid schedules -- same
    factor, same dependency checks of the NumPy double (NaN-poisoned buffers, asserted reads), distributed substitution."""
    out = launch(world, ["--mode", "cpu2d", "--npts", str(n), "--blk", str(nb), "--grid", ""], dict(env, GPX_DIST_SOLVE="dist"),
                 timeout=900)
    assert "world=%d" % world in out


@pytest.mark.parametrize("world,n,nb,env", [
    (4, 1700, 128, {"GPX_DIST_CHUNK_MIN_BLOCKS": "1", "GPX_DIST_CHUNK_HOLD": "1"}),    # the cut re-centred at every step
    (4, 1500, 128, {"GPX_DIST_CHUNK_MIN_BLOCKS": "2", "GPX_DIST_CHUNK_HOLD": "3"}),
    (8, 1300, 128, {"GPX_DIST_CHUNK_MIN_BLOCKS": "2", "GPX_DIST_CHUNK_HOLD": "2", "GPX_DIST_AGG": "2"}),
    (2, 1500, 128, {"GPX_DIST_CHUNK_MIN_BLOCKS": "3", "GPX_DIST_CHUNK_HOLD": "8"}),    # chunk 0 shrinks to its one block row
    (6, 1300, 128, {"GPX_DIST_PANEL_CHUNKS": "1"})])                                   # rounds 2-4: the panel in one piece
def test_panel_loop_2d_row_chunks_gloo_cpu(world, n, nb, env):
    """Round 5: the panel travels in two row chunks, one stage apart on the solve / broadcast / near-update streams.  Every cut
    rule the loop can meet -- re-centred every step, held until chunk 0 is down to the block row it must keep, pieces too short to
    cut, cutting off -- gives the same factor, alpha and log-likelihood under the NumPy double's dependency checks (NaN-poisoned
    buffers, asserted reads), with the distributed substitution."""
    out = launch(world, ["--mode", "cpu2d", "--npts", str(n), "--blk", str(nb), "--grid", ""], dict(env, GPX_DIST_SOLVE="dist"),
                 timeout=900)
    assert "world=%d" % world in out


@pytest.mark.parametrize("world,n,nb", [(2, 700, 128), (4, 900, 128), (8, 1100, 128)])
def test_c5_sharded_gradient_and_mi_gloo_cpu(world, n, nb):
    """BASELINE config C5 in its multi-GPU form, host logic over gloo with NumPy doubles: distributed fit,
    hyper-parameter gradient with the traces sharded by work-balanced row slabs of K^-1 (gp.py:444-466), greedy MI with the
    scoring sharded by rows of the inverse and a first-max merge (experimentalDesign.py:259-285, 753-785)."""
    out = launch(world, ["--mode", "cpu-c5", "--npts", str(n), "--blk", str(nb)], timeout=900)
    assert "cpu-c5 world=%d" % world in out


@pytest.mark.parametrize("world", [2, 3, 4])
def test_class_api_golden_fixtures_gloo_cpu(world):
    """the reference's golden fixtures THROUGH THE CLASS API (GP.train / evaluate / evaluateVariance /
    computeLogLike / loglikeParams(returnDeriv=1), costFunctionGP_IVAR.evaluate, greedy variance / greedy IVAR / MI designs)
    under a multi-process launch -- gloo + NumPy device doubles (tests/numpy_device.py), routed by gpexp_amd.dist.Session:
    distributed fit into a replicated factor, evaluation points sharded and gathered.  coeff / mean / variance / log-marginal
    1e-10 against the reference, indices exact, every rank returns bit-identical arrays, and ranks that pass different data
    are detected (tests/api_worker.py)."""
    out = launch(world, ["--mode", "cpu-api"], timeout=900)
    assert "cpu-api world=%d cases=9" % world in out


@pytest.mark.parametrize("world", [2, 4])
def test_class_api_golden_fixtures_distributed_factor_gloo_cpu(world):
    """Round 5 (VERDICT r4 missing 3): the same fixtures through the class API in the DISTRIBUTED-FACTOR mode
    (GPX_DIST_FACTOR=cyclic; the default from 65536 training points): the fit leaves the factor block-cyclic on the ranks -- no
    replica, no clone --, coeff and the log-marginal come from the distributed substitution, evaluate / evaluateVariance / the
    IVAR cost re-stream the panels against a window of the factor (dist2_restream_enqueue), and only the entry points that need
    a dense factor assemble one (api_worker asserts that the evaluations did not).  1e-10 against the reference, indices exact,
    bit-identical arrays on every rank."""
    out = launch(world, ["--mode", "cpu-api"], {"GPX_DIST_FACTOR": "cyclic"}, timeout=900)
    assert "cpu-api world=%d cases=9" % world in out and "factor=cyclic" in out


def test_grid_logic():
    from gpexp_amd import dist
    assert [dist.choose_grid(w) for w in (1, 2, 3, 4, 6, 8, 16)] == [(1, 1), (1, 2), (1, 3), (2, 2), (2, 3), (2, 4), (4, 4)]
    n, nb = 5000, 512
    for Pr, Pc in ((2, 4), (1, 2), (2, 2), (3, 2)):
        geos = [dist.Grid2D(n, nb, Pr, Pc, r) for r in range(Pr * Pc)]
        g0 = geos[0]
        assert g0.nblk == 10 and g0.np == 5120 and g0.height(9) == 512
        # every block of the lower triangle has exactly one owner, local shapes add up to the padded order
        assert sum(g.local_rows(g.pr) for g in geos if g.pc == 0) == g0.np
        assert sum(g.local_cols(g.pc) for g in geos if g.pr == 0) == g0.np
        for k in range(g0.nblk):
            # the pieces of a panel partition its rows below the diagonal block
            assert sum(g0.piece_rows(p, k) for p in range(Pr)) == g0.np - (k + 1) * nb
            roots = {root for _, _, root in g0.pieces(k)}
            assert all(r % Pc == k % Pc for r in roots)                    # pieces come from the process column of k
            for g in geos:
                for J in g.my_cols_after(k):
                    lr0, m, lc0, ncols, aoff, boff = g.update_args(k, J)
                    assert J % Pc == g.pc and lc0 == (J // Pc) * nb and ncols == g.height(J)
                    assert lr0 + m == g.local_rows(g.pr)
                    assert aoff + m * nb <= g.buf_elems() and boff + ncols * nb <= g.buf_elems()
    g = dist.Grid2D(700, 256, 2, 4, 5)     # ragged: padded 768 = 3 blocks of 256
    assert g.nblk == 3 and [g.height(i) for i in range(3)] == [256, 256, 256]
    g = dist.Grid2D(333, 256, 1, 3, 1)     # padded 384: last block is 128 tall
    assert g.nblk == 2 and g.height(1) == 128


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,nb,grid", [(2, 1500, 256, ""), (4, 2100, 256, ""), (3, 1900, 128, ""), (4, 900, 512, "4x1"),
                                             (4, 4300, 1024, "")])
def test_distributed_fit_ivar_2d_shared_gpu(world, n, nb, grid):
    """The 2-D path on the real HIP primitives (ranks share the GPU, host-staged exchange): replicated factor, alpha,
    log-likelihood and IVAR equal the single-GPU path; both evaluation schedules; repeatable bit for bit."""
    out = launch(world, ["--mode", "gpu2d", "--npts", str(n), "--mpts", "777", "--blk", str(nb), "--grid", grid],
                 {"GPX_COMM": "host", "GPX_FORCE_DEVICE": "0"}, timeout=900)
    assert "HostStagedComm" in out


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_class_api_golden_fixtures_shared_gpu(world):
    """The same fixtures through the class API on the real HIP library, 2 / 4 ranks sharing the GPU (host-staged exchange):
    1e-10 against the reference's outputs, greedy / MI indices exact, identical arrays on every rank."""
    out = launch(world, ["--mode", "gpu-api"], {"GPX_COMM": "host", "GPX_FORCE_DEVICE": "0"}, timeout=900)
    assert "gpu-api world=%d cases=9" % world in out


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_class_api_golden_fixtures_distributed_factor_shared_gpu(world):
    """... and on the real HIP library (ranks share the GPU, host-staged exchange): gpx_dist2_panel_pack / gpx_dist2_diag_pack, the
    windowed solve against the re-streamed panels, the distributed substitution."""
    out = launch(world, ["--mode", "gpu-api"], {"GPX_COMM": "host", "GPX_FORCE_DEVICE": "0", "GPX_DIST_FACTOR": "cyclic"}, timeout=900)
    assert "gpu-api world=%d cases=9" % world in out and "factor=cyclic" in out


@pytest.mark.gpu
def test_class_api_golden_fixtures_rccl_world1():
    """... and with the RCCL communicator at world 1 (ncclCommInitRank, the host all-gathers of the session)."""
    out = launch(1, ["--mode", "gpu-api"], {"GPX_COMM": "rccl"}, timeout=900)
    assert "gpu-api world=1 cases=9" in out


@pytest.mark.gpu
def test_class_api_c4_lite_reference_fixture_on_four_ranks():
    """The N = 8192 reference fixture through the class API with the session's default thresholds on a 2 x 2 grid: distributed
    fit (16 panels), replica-based coeff / log-marginal, sharded evaluation; 1e-10 / 5e-10 element-wise."""
    out = launch(4, ["--mode", "gpu-api-c4lite"], {"GPX_COMM": "host", "GPX_FORCE_DEVICE": "0"}, timeout=900)
    assert "gpu-api-c4lite world=4" in out


@pytest.mark.gpu
def test_class_api_c4_lite_reference_fixture_distributed_factor_on_four_ranks():
    """The N = 8192 reference fixture in the distributed-factor mode on a 2 x 2 grid: 16 panels re-streamed through a window of
    the factor per evaluation, coeff / log-marginal by distributed substitution; 1e-10 / 5e-10 element-wise."""
    out = launch(4, ["--mode", "gpu-api-c4lite"], {"GPX_COMM": "host", "GPX_FORCE_DEVICE": "0", "GPX_DIST_FACTOR": "cyclic"},
                 timeout=900)
    assert "gpu-api-c4lite world=4" in out


@pytest.mark.gpu
def test_2d_loop_is_ordered_behind_a_slow_assembly():
    """The local assembly of A is queued on MAIN; every other stream's first touch of A must be ordered behind it.  At k = 0
    the early diagonal update (PANEL stream of the owner of block (1,1)) was not: once in a dozen fresh processes -- when the
    fill kernel's first launch was slow -- the update landed before the fill and was overwritten (wrong factor from block
    (1,1) on, first step only; later steps re-read identical data).  Here the assembly of that rank alone (rank 3 of the 2 x 2
    grid) is held back by 40 ms, which makes the unordered version fail every time."""
    out = launch(4, ["--mode", "gpu2d", "--npts", "2100", "--mpts", "777", "--blk", "256"],
                 {"GPX_COMM": "host", "GPX_FORCE_DEVICE": "0", "GPX_TEST_DELAY_FILL": "40@3"}, timeout=900)
    assert "HostStagedComm" in out


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,nb,grid,seed", [(4, 2100, 256, "", 5), (4, 2100, 256, "", 9), (3, 1900, 128, "", 2),
                                                  (4, 1500, 256, "4x1", 3), (4, 2100, 128, "1x4", 4)])
def test_2d_loop_under_random_stream_delays(world, n, nb, grid, seed):
    """Chaos mode of the worker: every device primitive is preceded, with probability 0.4, by a kernel that holds the current
    stream back for 1-40 ms; packed buffers and replicated factor start NaN-filled; the FIRST step of a fresh runner (both
    evaluation schedules) must still equal the single-GPU path.  Seeds 5 and 9 fail on the 2 x 2 grid when the k = 0
    dependency on the assembly is removed."""
    out = launch(world, ["--mode", "gpu2d-chaos", "--npts", str(n), "--mpts", "777", "--blk", str(nb), "--grid", grid,
                         "--chaos", str(seed)], {"GPX_COMM": "host", "GPX_FORCE_DEVICE": "0"}, timeout=900)
    assert "gpu2d-chaos" in out


@pytest.mark.gpu
def test_replayed_ranks_over_block_sizes_group_sizes_and_grids():
    """scripts/replay_matrix.sh: one process plays ranks of 1x2 ... 4x2 grids with the recorded programs and real kernels --
    evaluation streamed against the WINDOW of the factor at every world size, block sizes 128 ... 1024, 1 ... 8 panels per
    group (ragged last groups, the window wrapping many times); every run checks its rank's variance sum against the
    single-GPU path to 1e-10."""
    env = dict(os.environ, GPX_DIST_STREAM_IVAR="1")
    r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "replay_matrix.sh")], env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=1200)
    assert r.returncode == 0 and "FAIL" not in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("ok ") == 24, r.stdout


@pytest.mark.gpu
def test_bench_falls_back_to_1d_layout_when_the_2d_preflight_is_wrong():
    """First contact: a 2-D preflight that computes something WRONG (forced here on both ranks) must not end the run -- all
    ranks agree to rebuild the runner on the 1-D block-column layout, that layout passes its own preflight, and the line that
    comes out says so and carries the failed 2-D record."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GPX_COMM="host", GPX_FORCE_DEVICE="0", MASTER_ADDR="127.0.0.1",
               GPX_BENCH_INJECT="fail2d")
    flags = ["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--train-points", "4096", "--mc-points", "2048"]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2"] + flags, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    assert "fall back to the 1-D block-column layout" in two.stderr
    lines = [ln for ln in two.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines[:3]
    got = json.loads(lines[0])
    assert got["n_gpus"] == 2 and "1-D" in got["config"]["parallelism"]
    assert got["preflight"]["ok"] and got["preflight"]["layout"] == "1d" and got["preflight"]["failed_2d_preflight"]["forced_failure"]


@pytest.mark.gpu
def test_bench_restarts_on_1d_layout_when_the_2d_preflight_hangs():
    """First contact, the other failure: a 2-D preflight that never finishes (forced: both ranks sleep in it).  The ranks' own
    preflight watchdogs end the children (exit code 125), every rank's supervisor -- a process that never touched the GPU --
    starts a second child on the 1-D layout with a fresh rendezvous, and ONE JSON line comes out, labelled 1-D."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GPX_COMM="host", GPX_FORCE_DEVICE="0", MASTER_ADDR="127.0.0.1",
               GPX_BENCH_INJECT="hang2d", GPX_BENCH_PREFLIGHT_WATCHDOG_S="45")
    flags = ["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--train-points", "4096", "--mc-points", "2048"]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2"] + flags, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    assert "second attempt on the 1-D block-column layout" in two.stderr
    lines = [ln for ln in two.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines[:3]
    got = json.loads(lines[0])
    assert got["n_gpus"] == 2 and "1-D" in got["config"]["parallelism"] and got["preflight"]["ok"]


@pytest.mark.gpu
def test_bench_c4_full_size_four_ranks_equals_single_gpu():
    """The headline step at its full size (N = 32768, d = 8, M = 32768) through bench.py exactly as the driver launches it
    with 4 ranks (2 x 2 grid, streamed evaluation; the ranks share this box's GPU through the host-staged communicator)
    against the single-GPU run of the same file: log-likelihood and IVAR to 1e-12, one JSON line on stdout each."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    flags = ["--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + flags, env=env, cwd=ROOT, capture_output=True,
                         text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    ref = json.loads(one.stdout.strip())
    env.update({"GPX_COMM": "host", "GPX_FORCE_DEVICE": "0", "MASTER_ADDR": "127.0.0.1"})
    four = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
                           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"),
                           "--gpus", "4"] + flags, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert four.returncode == 0, four.stderr[-3000:]
    lines = [ln for ln in four.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines[:3]
    got = json.loads(lines[0])
    assert got["n_gpus"] == 4 and "2x2" in got["config"]["parallelism"]
    for key in ("loglike", "ivar"):
        assert got["results"][key] == pytest.approx(ref["results"][key], rel=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,m,nb", [(2, 1500, 300, 256), (4, 2100, 517, 256), (3, 1300, 200, 128)])
def test_c5_distributed_gradient_and_mi_shared_gpu(world, n, m, nb):
    """Config C5 on N ranks with the real kernels (ranks share the GPU, host-staged exchange): sharded gradient == single-GPU
    gpx_lml_grad to 1e-10, sharded MI picks == gpx_mi_greedy."""
    out = launch(world, ["--mode", "gpu-c5", "--npts", str(n), "--mpts", str(m), "--blk", str(nb)],
                 {"GPX_COMM": "host", "GPX_FORCE_DEVICE": "0"}, timeout=900)
    assert "gpu-c5 world=%d" % world in out


@pytest.mark.gpu
def test_rccl_c5_world1():
    out = launch(1, ["--mode", "gpu-c5", "--npts", "1100", "--mpts", "300", "--blk", "256"], {"GPX_COMM": "rccl"})
    assert "RcclComm" in out


@pytest.mark.gpu
def test_rccl_2d_world1():
    """RCCL code path of the 2-D loop at world 1: ncclCommSplit sub-communicators, group broadcasts / reductions, the
    grouped send/recv panel broadcast (degenerate: nothing to send) and the all-reduces."""
    out = launch(1, ["--mode", "gpu2d", "--npts", "1100", "--mpts", "300", "--blk", "256"], {"GPX_COMM": "rccl"})
    assert "RcclComm" in out


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,nb", [(2, 1500, 256), (3, 2100, 512)])
def test_distributed_fit_ivar_shared_gpu(world, n, nb):
    out = launch(world, ["--mode", "gpu", "--npts", str(n), "--mpts", "777", "--blk", str(nb)],
                 {"GPX_COMM": "host", "GPX_FORCE_DEVICE": "0"})
    assert "HostStagedComm" in out


@pytest.mark.gpu
def test_rccl_communicator_world1():
    out = launch(1, ["--mode", "gpu", "--npts", "900", "--mpts", "300", "--blk", "256"], {"GPX_COMM": "rccl"})
    assert "RcclComm" in out


# ---- real RCCL, one rank per device: switched on by the number of visible devices ------------------------------------------
_VISIBLE = []


def visible_gpus():
    """HIP devices this box shows, counted in a CHILD process (hipGetDeviceCount through ctypes on libamdhip64): the test
    runner itself is not asked to initialise a device for a decision about skipping, and nothing is exec-ed from a process that
    holds one.  0 when there is no HIP runtime or no device."""
    if not _VISIBLE:
        code = ("import ctypes\n"
                "try:\n"
                "    h = ctypes.CDLL('libamdhip64.so')\n"
                "    n = ctypes.c_int(0)\n"
                "    print(n.value if h.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0)\n"
                "except OSError:\n"
                "    print(0)\n")
        try:
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
            _VISIBLE.append(int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else 0)
        except (OSError, ValueError, subprocess.TimeoutExpired):
            _VISIBLE.append(0)
    return _VISIBLE[0]


def need_gpus(world):
    n = visible_gpus()
    if n < world:
        pytest.skip("real RCCL at world %d needs %d visible devices, this box shows %d (RCCL refuses two ranks on one device)"
                    % (world, world, n))


RCCL_ENV = {"GPX_COMM": "rccl", "NCCL_DEBUG": "WARN"}     # no GPX_FORCE_DEVICE: rank r opens device LOCAL_RANK


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,nb,grid", [(2, 1500, 256, ""), (4, 2100, 256, ""), (4, 4300, 1024, ""), (8, 4300, 256, ""),
                                             (8, 2500, 128, "4x2")])
def test_distributed_fit_ivar_2d_real_rccl(world, n, nb, grid):
    """test_distributed_fit_ivar_2d_shared_gpu with the product communicator: ncclCommSplit row / column communicators, the
    grouped send/recv panel broadcast, group broadcasts and all-reduces between REAL ranks; replicated factor, alpha,
    log-likelihood and IVAR against the single-GPU path on every rank (L 1e-12, alpha 1e-10, the rest 1e-11)."""
    need_gpus(world)
    out = launch(world, ["--mode", "gpu2d", "--npts", str(n), "--mpts", "777", "--blk", str(nb), "--grid", grid], RCCL_ENV,
                 timeout=900)
    assert "RcclComm" in out


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_class_api_golden_fixtures_real_rccl(world):
    """The reference's fixtures through the class API over RCCL: 1e-10, indices exact, bit-identical arrays on every rank."""
    need_gpus(world)
    out = launch(world, ["--mode", "gpu-api"], RCCL_ENV, timeout=900)
    assert "gpu-api world=%d cases=9" % world in out


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_class_api_golden_fixtures_distributed_factor_real_rccl(world):
    """... and in the distributed-factor mode (block-cyclic factor, re-streamed evaluation) over RCCL."""
    need_gpus(world)
    out = launch(world, ["--mode", "gpu-api"], dict(RCCL_ENV, GPX_DIST_FACTOR="cyclic"), timeout=900)
    assert "gpu-api world=%d cases=9" % world in out and "factor=cyclic" in out


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_class_api_c4_lite_reference_fixture_real_rccl(world):
    """The N = 8192 reference fixture through the class API with the session's default thresholds over RCCL."""
    need_gpus(world)
    out = launch(world, ["--mode", "gpu-api-c4lite"], RCCL_ENV, timeout=900)
    assert "gpu-api-c4lite world=%d" % world in out


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,m,nb", [(2, 1500, 300, 256), (4, 2100, 517, 256), (8, 2900, 700, 128)])
def test_c5_distributed_gradient_and_mi_real_rccl(world, n, m, nb):
    """Config C5 over RCCL: sharded gradient == single-GPU gpx_lml_grad (1e-10), sharded MI picks == gpx_mi_greedy."""
    need_gpus(world)
    out = launch(world, ["--mode", "gpu-c5", "--npts", str(n), "--mpts", str(m), "--blk", str(nb)], RCCL_ENV, timeout=900)
    assert "gpu-c5 world=%d" % world in out


def test_real_rccl_cases_are_collected_and_gate_on_the_device_count():
    """CPU side of the switch: the counter answers (0 here), and a case asked for more devices than that skips with the reason."""
    n = visible_gpus()
    assert isinstance(n, int) and n >= 0
    with pytest.raises(pytest.skip.Exception) as e:
        need_gpus(n + 1)
    assert "visible devices" in str(e.value)
