"""Runs last (file name): with GPX_ALLOC_GUARD=1 in the environment every pooled device allocation of the session carried
guard bands; nothing may have written outside its block.  (How the scratch overflow of the column reduction would have been
seen; the pool's GPU boxes have no address sanitizer.)  Usage: GPX_ALLOC_GUARD=1 python -m pytest tests -m gpu   (=2: blocks are also handed out NaN-filled)"""
import os

import pytest

pytestmark = pytest.mark.gpu


def test_no_allocation_guard_was_violated():
    from gpexp_amd import device as dev
    ctx = dev.context()
    ctx.sync()
    ctx.trim()                       # blocks still in flight are checked when they return; trim only drops the pool
    n = ctx.guard_violations()
    if os.environ.get("GPX_ALLOC_GUARD", "0") in ("", "0"):
        assert n == -1
        pytest.skip("guard mode off (set GPX_ALLOC_GUARD=1)")
    assert n == 0, "%d pooled blocks were overwritten outside their bounds (see stderr)" % n
    assert ctx.lib.gpx_dbg_guard_selftest(ctx.h) == 1   # the check itself works: a deliberate 16-byte overrun is caught
