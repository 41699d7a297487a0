"""The one-launch GroupNorm backward (groupnorm.hip: gn_bwd_coop_kernel) is a pipeline of work-groups that meet through counters.
This file replays its stage program on the CPU from the library's own plan (``mas_gn_bwd_plan``, host arithmetic) under random
work-group interleavings and checks what the GPU tests cannot see directly:

* the schedule cannot deadlock (every wait is for arrivals that are due strictly earlier in every work-group's program order);
* a slot of the row ring is never overwritten before the finalize task that reads it has run, for every pipeline depth;
* every (image, pixel row) is reduced exactly once and applied exactly once, by the same work-group, after its coefficients exist;
* the finalize tasks of a group go to distinct work-groups when there are enough of them.

No kernel runs here (no GPU): the model mirrors the loop structure of the kernel line by line -- S(t), F(t-1), P(t-D)."""
import ctypes
import random

import pytest


def _plan(n, hw, c, g=32, cus=256):
    import mas_hip
    out = (ctypes.c_int * 10)()
    ok = mas_hip.lib().mas_gn_bwd_plan(n, hw, c, g, cus, out)
    keys = ("T", "W", "nsplit", "Gi", "NG", "NS", "CS", "D", "R", "mult")
    return bool(ok), dict(zip(keys, out))


SHAPES = [(32, 256 * 256, 128), (32, 128 * 128, 128), (32, 128 * 128, 256), (32, 64 * 64, 256), (32, 64 * 64, 512), (32, 32 * 32, 512),
          (32, 16 * 16, 512), (32, 64 * 64, 128), (2, 32 * 32, 32), (3, 12 * 20, 64), (1, 9 * 13, 128), (24, 128 * 128, 128), (7, 64 * 64, 256),
          (4, 8 * 8, 32)]


@pytest.mark.parametrize("shape", SHAPES)
def test_plan_invariants(shape):
    n, hw, c = shape
    ok, p = _plan(n, hw, c)
    assert ok, shape
    upp, cpg = c // 8, c // 32
    assert p["T"] in (256, 512, 1024) and p["T"] % upp == 0
    assert 1 <= p["W"] <= 4 * 256 and p["W"] * p["T"] <= 256 * 1024          # at most half of the chip's thread slots
    assert 1 <= p["nsplit"] <= min(p["W"], hw)
    assert p["Gi"] * p["nsplit"] <= p["W"] or p["Gi"] == 1
    assert p["NG"] == -(-n // p["Gi"])
    assert p["CS"] * p["NS"] == c and p["CS"] % cpg == 0 and p["CS"] <= p["T"]
    assert 1 <= p["D"] <= 4 and p["R"] == p["D"] + 2
    from math import gcd
    assert gcd(p["mult"], p["W"]) == 1
    # workspace bound used by mas_gn_bwd_workspace: ring rows <= 6 * 4 * CUs
    assert p["R"] * p["Gi"] * p["nsplit"] <= 6 * 4 * 256


def test_benched_shape_plan():
    """32 x 128 ch x 256^2 on 256 CUs: one image per group (33.5 MB of x + da), 32 groups, every work-group one row range"""
    ok, p = _plan(32, 65536, 128)
    assert ok and p["Gi"] == 1 and p["NG"] == 32 and p["nsplit"] == p["W"] == 512 and p["T"] == 512
    assert p["NS"] * p["CS"] == 128 and p["nsplit"] * p["CS"] * 8 <= 65536


def _simulate(n, hw, c, plan, seed, depth=None):
    p = dict(plan)
    if depth is not None:
        p["D"], p["R"] = depth, depth + 2
    W, NG, Gi, nsplit, NS, D, R, mult = p["W"], p["NG"], p["Gi"], p["nsplit"], p["NS"], p["D"], p["R"], p["mult"]
    rows_per = -(-hw // nsplit)
    rev = True
    group_of = (lambda t: NG - 1 - t) if rev else (lambda t: t)
    cntA, cntB = [0] * NG, [0] * NG
    ring = {}                     # (slot, row) -> (t, state) with state in {"written", "read"}
    coef = set()                  # (image, slice) finalized
    reduced, applied = {}, {}     # (image, row range) -> work-group
    rng = random.Random(seed)

    def program(w):
        for t in range(NG + D):
            if t < NG:
                g = group_of(t); n0 = g * Gi; cn = min(Gi, n - n0)
                for pr in range(w, cn * nsplit, W):
                    i, sp = divmod(pr, nsplit)
                    r0 = min(hw, sp * rows_per); r1 = min(hw, r0 + rows_per)
                    key = (t % R, i * nsplit + sp)
                    prev = ring.get(key)
                    assert prev is None or prev[1] == "read", f"ring slot {key} of stage {prev[0]} overwritten at stage {t} before it was read"
                    ring[key] = (t, "written")
                    assert (n0 + i, r0, r1) not in reduced
                    reduced[(n0 + i, r0, r1)] = w
                cntA[t] += 1
                yield None
            if 1 <= t and t - 1 < NG:
                tf = t - 1; g = group_of(tf); n0 = g * Gi; cn = min(Gi, n - n0)
                waited = False
                for k in range(cn * NS):
                    if ((tf * Gi * NS + k) * mult) % W != w:
                        continue
                    if not waited:
                        while cntA[tf] < W:
                            yield ("A", tf)
                        waited = True
                    i, sl = divmod(k, NS)
                    for r in range(nsplit):
                        key = (tf % R, i * nsplit + r)
                        assert ring[key][0] == tf, f"finalize of stage {tf} reads a row of stage {ring[key][0]}"
                    coef.add((n0 + i, sl))
                    cntB[tf] += 1
                    yield None
            if D <= t and t - D < NG:
                ta = t - D; g = group_of(ta); n0 = g * Gi; cn = min(Gi, n - n0)
                while cntB[ta] < cn * NS:
                    yield ("B", ta)
                # every task of the group has read its rows: the slot may be reused
                for i in range(cn):
                    for r in range(nsplit):
                        key = (ta % R, i * nsplit + r)
                        if ring[key][0] == ta:
                            ring[key] = (ta, "read")
                for pr in range(w, cn * nsplit, W):
                    i, sp = divmod(pr, nsplit)
                    r0 = min(hw, sp * rows_per); r1 = min(hw, r0 + rows_per)
                    assert all((n0 + i, sl) in coef for sl in range(NS))
                    assert reduced[(n0 + i, r0, r1)] == w                    # the same work-group re-reads what it reduced (L2 / MALL locality)
                    assert (n0 + i, r0, r1) not in applied
                    applied[(n0 + i, r0, r1)] = w
                yield None

    gens = {w: program(w) for w in range(W)}
    blocked = {}
    steps = 0
    while gens:
        runnable = [w for w in gens if w not in blocked or
                    (cntA if blocked[w][0] == "A" else cntB)[blocked[w][1]] >= (W if blocked[w][0] == "A" else min(Gi, n - group_of(blocked[w][1]) * Gi) * NS)]
        assert runnable, f"deadlock: {len(gens)} work-groups blocked on {set(blocked.values())}"
        w = rng.choice(runnable)
        blocked.pop(w, None)
        try:
            r = next(gens[w])
            if r is not None:
                blocked[w] = r
        except StopIteration:
            del gens[w]
        steps += 1
        assert steps < 5_000_000
    # coverage: every pixel row of every image exactly once, in both phases
    for img in range(n):
        spans = sorted((r0, r1) for (i, r0, r1) in reduced if i == img and r1 > r0)
        assert spans[0][0] == 0 and spans[-1][1] == hw and all(a[1] == b[0] for a, b in zip(spans, spans[1:])), (img, spans[:4])
    assert set(reduced) == set(applied)
    assert len(coef) == n * NS


@pytest.mark.parametrize("shape", [(6, 32 * 32, 64), (9, 16 * 16, 128), (40, 24 * 24, 32), (12, 64 * 64, 128), (30, 32 * 32, 256)])
@pytest.mark.parametrize("depth", [1, 2, 3, 4])
def test_schedule_is_deadlock_free_and_ring_safe(shape, depth):
    """a small device (8 CUs -> few work-groups, several groups) so that the ring wraps many times; three random interleavings"""
    n, hw, c = shape
    ok, p = _plan(n, hw, c, cus=8)
    assert ok
    assert p["NG"] >= 2
    for seed in range(3):
        _simulate(n, hw, c, p, seed, depth)


def test_finalize_tasks_of_a_group_have_distinct_owners():
    for shape in SHAPES:
        ok, p = _plan(*shape)
        tpg = p["Gi"] * p["NS"]
        if tpg > p["W"]:
            continue
        for tf in range(p["NG"]):
            owners = [((tf * tpg + k) * p["mult"]) % p["W"] for k in range(tpg)]
            assert len(set(owners)) == tpg, (shape, tf)
