"""Every `GPX_*` environment switch the product reads (at most 30, each documented in README.md and exercised
by a test with a non-default value).  This file holds the tests of the switches no other test file touches, and the census that
keeps the README table and the product tree in step."""
import os
import re
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRODUCT = os.path.join(ROOT, "gpexp_amd")


def product_switches():
    names = set()
    for base, _, files in os.walk(PRODUCT):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                with open(os.path.join(base, f)) as fh:
                    names.update(re.findall(r'"(GPX_[A-Z0-9_]+)"', fh.read()))
    return names


def readme_table():
    """{switch: (default, test)} from the table under '## Environment switches' in README.md."""
    rows = {}
    with open(os.path.join(ROOT, "README.md")) as f:
        text = f.read().split("## Environment switches", 1)[1]
    for line in text.splitlines():
        m = re.match(r"\|\s*`(GPX_[A-Z0-9_]+)`\s*\|([^|]*)\|([^|]*)\|([^|]*)\|", line)
        if m:
            rows[m.group(1)] = (m.group(2).strip(), m.group(4).strip())
    return rows


def test_switch_census_matches_readme():
    """At most 30 switches are read under gpexp_amd/, README.md documents every one with its default and names the test that
    runs it with a non-default value -- and that test file really mentions the switch."""
    names = product_switches()
    table = readme_table()
    assert len(names) <= 30, sorted(names)
    assert names == set(table), (sorted(names - set(table)), sorted(set(table) - names))
    for name, (default, test) in table.items():
        assert default, name
        path = os.path.join(ROOT, "tests", test.strip("`").split("::")[0])
        assert os.path.exists(path), (name, test)
        with open(path) as f:
            assert name in f.read(), (name, test)


def test_lib_path_switch_loads_the_named_library(tmp_path):
    """GPX_LIB_PATH: an alternative build of the same ABI (A/B tests of a kernel change against the committed tree)."""
    src = os.path.join(PRODUCT, "libgpx_hip.so")
    if not os.path.exists(src):
        pytest.skip("library not built")
    alt = str(tmp_path / "libgpx_alt.so")
    shutil.copy(src, alt)
    code = ("import sys; sys.path.insert(0, %r)\nfrom gpexp_amd import _lib\nlib = _lib.load()\n"
            "print(lib._name, lib.gpx_abi_version())\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GPX_LIB_PATH=alt), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split() == [alt, "2"]
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GPX_LIB_PATH=alt + ".missing"), capture_output=True, text=True,
                       timeout=120)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr


def test_rdv_key_names_the_rendezvous(tmp_path):
    """GPX_RDV_KEY: ranks that do not share a parent process (bench.py's supervised children) meet under an explicit key."""
    code = ("import sys; sys.path.insert(0, %r)\nfrom gpexp_amd import dist\nr = dist.FileRendezvous()\n"
            "print(r.path, flush=True)\ngot = r.exchange(b'k' * 8 if r.rank == 0 else b'?' * 8)\nprint(got.decode(), flush=True)\nimport time; time.sleep(1.0 if r.rank == 0 else 0.0)\nr.cleanup()\n"
            % ROOT)
    with open(os.path.join(PRODUCT, "dist.py")) as f:
        if "class FileRendezvous" not in f.read():
            pytest.skip("rendezvous class renamed")
    env = dict(os.environ, WORLD_SIZE="2", MASTER_PORT="45125", GPX_RDV_DIR=str(tmp_path), GPX_RDV_KEY="explicit")
    procs = [subprocess.Popen([sys.executable, "-c", code], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, text=True) for r in (1, 0)]
    outs = [p.communicate(timeout=120)[0].split() for p in procs]
    assert all(p.returncode == 0 for p in procs)
    assert all(o[0].endswith("gpx_rdv_explicit") and o[1] == "kkkkkkkk" for o in outs), outs


def test_dist_nb_switch_overrides_the_block_size(monkeypatch):
    """GPX_DIST_NB: block size of the 2-D block-cyclic layout (default: 1024 for the factorisation on 8 ranks from N = 16384, else
    512 / 256 / 128 by N) -- the paced-replay sweeps and bench.py's 1-D fall-back layout set it."""
    from gpexp_amd import dist
    assert dist.default_nb(32768, 8) == 1024 and dist.default_nb(32768, 4) == 512 and dist.default_nb(4096, 8) == 256
    monkeypatch.setenv("GPX_DIST_NB", "2048")
    assert dist.default_nb(32768, 8) == 2048 and dist.default_nb(1000, 2) == 2048


CHILD = r"""
import sys, os
sys.path.insert(0, %r)
import numpy as np
from gpexp_amd import device as dev
ctx = dev.context()
rng = np.random.default_rng(8192)
n, d = 8192, 6
X = dev.points(ctx, rng.uniform(-1, 1, (n, d)))
Z = dev.points(ctx, rng.uniform(-1, 1, (700, d)))
y = rng.standard_normal(n)
sp = dev.KernelSpec(dev.K_MATERN52, d, [0.6, 1.0])
K = dev.potrf(ctx, dev.kfill(ctx, sp, X, nugget=0.05))
a = dev.potrs(ctx, K, y)
print("RESULT %%r %%r %%r" %% (dev.logdet(ctx, K), float(y @ a), dev.ivar(ctx, sp, K, X, Z)), flush=True)
"""


def run_child(extra):
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=dict(os.environ, **extra), cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0]
    return line, r.stderr


@pytest.mark.gpu
def test_chaos_delays_change_nothing_and_potrf_timing_reports_panels():
    """GPX_CHAOS=<seed>: every launch site holds its stream back at random -- a fit (blocked look-ahead factorisation: two
    streams), a solve and an IVAR come out bit-identical, i.e. no dependency between the context's streams is met by lucky
    timing.  GPX_POTRF_TIMING=1: the per-panel phase spans of the blocked factorisation on stderr."""
    plain, _ = run_child({})
    for seed in ("5", "77"):
        chaotic, _ = run_child({"GPX_CHAOS": seed})
        assert chaotic == plain
    timed, err = run_child({"GPX_POTRF_TIMING": "1"})
    assert timed == plain
    assert re.search(r"potrf-timing panel 0 top\s+at\s+[-0-9.]+ ms\s+len\s+[0-9.]+ ms", err) and "potrf-timing panel 0 chain" in err


@pytest.mark.gpu
def test_event_timing_switch_makes_pipeline_events_carry_time_stamps():
    """GPX_EVENT_TIMING=1 (bench.py's multi_gpu_replay reads the distributed loop's strands this way): gpx_dbg_event_elapsed
    between two recorded pipeline events around a 3 ms spin; without the switch the events carry no time."""
    code = ("import sys, ctypes as C; sys.path.insert(0, %r)\nfrom gpexp_amd import device as dev\nctx = dev.context()\n"
            "ctx.record(0)\nctx.lib.gpx_dbg_spin(ctx.h, 3)\nctx.record(1)\nctx.sync()\nms = C.c_double(-1.0)\n"
            "rc = ctx.lib.gpx_dbg_event_elapsed(ctx.h, 0, 1, C.byref(ms))\nprint('EV', rc, ms.value, flush=True)\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GPX_EVENT_TIMING="1"), cwd=ROOT, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    _, rc, ms = [l for l in r.stdout.splitlines() if l.startswith("EV")][0].split()
    assert int(rc) == 0 and 2.5 <= float(ms) <= 50.0
    env = {k: v for k, v in os.environ.items() if k != "GPX_EVENT_TIMING"}
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and int([l for l in r.stdout.splitlines() if l.startswith("EV")][0].split()[1]) != 0
