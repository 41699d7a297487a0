"""The one-launch GroupNorm backward (groupnorm.hip: gn_bwd_queue_kernel) is a queue of reduce / finalize / apply tasks handed out by
one ticket counter to however many work-groups are resident.  This file replays the queue on the CPU from the library's own plan
(``mas_gn_bwd_plan``, host arithmetic) under random work-group interleavings and checks what the GPU tests cannot see directly:

* the queue cannot deadlock for ANY number of resident work-groups, one included (every wait is for tasks with smaller tickets,
  and the smallest unfinished ticket never waits);
* a slot of the row ring is never overwritten before every finalize task that reads it has run, for every lead;
* every (image, pixel row) is reduced exactly once and applied exactly once, after its coefficients exist, and the parameter-gradient
  tasks run after every group is finalized.

No kernel runs here (no GPU): the model mirrors the kernel's ticket decode and its three task bodies line by line."""
import ctypes
import random

import pytest

KEYS = ("T", "W", "nsplit", "Gi", "NG", "NS", "CS", "L", "R", "tpr")


def _plan(n, hw, c, g=32, cus=256):
    import mas_hip
    out = (ctypes.c_int * 10)()
    ok = mas_hip.lib().mas_gn_bwd_plan(n, hw, c, g, cus, out)
    return bool(ok), dict(zip(KEYS, out))


SHAPES = [(32, 256 * 256, 128), (32, 128 * 128, 128), (32, 128 * 128, 256), (32, 64 * 64, 256), (32, 64 * 64, 512), (32, 32 * 32, 512),
          (32, 16 * 16, 512), (32, 64 * 64, 128), (2, 32 * 32, 32), (3, 12 * 20, 64), (1, 9 * 13, 128), (24, 128 * 128, 128), (7, 64 * 64, 256),
          (4, 8 * 8, 32), (192, 256 * 256, 128)]


@pytest.mark.parametrize("shape", SHAPES)
def test_plan_invariants(shape):
    n, hw, c = shape
    ok, p = _plan(n, hw, c)
    assert ok, shape
    upp, cpg = c // 8, c // 32
    assert p["T"] in (256, 512) and p["T"] % upp == 0
    assert 1 <= p["W"] <= 8 * 256
    assert 1 <= p["nsplit"] <= min(512, hw)
    assert 1 <= p["Gi"] <= n and p["NG"] == -(-n // p["Gi"])
    assert p["CS"] * p["NS"] == c and p["CS"] % cpg == 0 and p["CS"] <= p["T"]
    assert 1 <= p["L"] <= 4 and p["R"] == p["L"] + 2
    assert p["tpr"] == p["Gi"] * p["NS"] + 2 * p["Gi"] * p["nsplit"]
    # workspace bound used by mas_gn_bwd_workspace: ring rows per group <= 1024, <= 6 slots
    assert p["Gi"] * p["nsplit"] <= 1024 and p["R"] <= 6


def test_benched_shape_plan():
    """32 x 128 ch x 256^2 on 256 CUs: one image per group (33.5 MB of x + da), 32 groups, 512 row ranges of 64 KiB per tensor"""
    ok, p = _plan(32, 65536, 128)
    assert ok and p["Gi"] == 1 and p["NG"] == 32 and p["nsplit"] == 512 and p["T"] == 512 and p["W"] == 512
    assert p["NS"] * p["CS"] == 128 and p["nsplit"] * p["CS"] * 8 <= 65536


def _simulate(n, hw, c, plan, resident, seed, lead=None):
    p = dict(plan)
    if lead is not None:
        p["L"], p["R"] = lead, lead + 2
    NG, Gi, nsplit, NS, L, R = p["NG"], p["Gi"], p["nsplit"], p["NS"], p["L"], p["R"]
    rows_per = -(-hw // nsplit)
    nF, nS = Gi * NS, Gi * nsplit
    tpr = nF + 2 * nS
    n_main = (NG + L) * tpr
    n_tail = -(-c // p["T"])
    group_of = lambda t: NG - 1 - t                       # (the kernel walks the batch back to front)
    images_of = lambda t: min(Gi, n - group_of(t) * Gi)
    head = [0]
    cntA, cntB = [0] * NG, [0] * NG
    ring = {}                                             # (slot, row) -> group whose sums it holds
    coef, reduced, applied, tails = set(), set(), set(), []
    rng = random.Random(seed)

    def worker():
        while True:
            tk = head[0]; head[0] += 1
            if tk >= n_main + n_tail:
                return
            if tk >= n_main:
                for t in range(NG):
                    while cntB[t] < images_of(t) * NS:
                        yield ("B", t, tk)
                assert len(coef) == n * NS
                tails.append(tk - n_main)
                continue
            rnd, idx = divmod(tk, tpr)
            if idx < nF:
                t = rnd - 1
                if t < 0 or t >= NG:
                    continue
                cn, n0 = images_of(t), group_of(t) * Gi
                i, sl = divmod(idx, NS)
                if i >= cn:
                    continue
                while cntA[t] < cn * nsplit:
                    yield ("A", t, tk)
                for r in range(nsplit):
                    assert ring[(t % R, i * nsplit + r)] == t, f"finalize of group {t} reads a row of group {ring[(t % R, i * nsplit + r)]}"
                coef.add((n0 + i, sl))
                cntB[t] += 1
                yield None
            elif idx < nF + nS:
                t = rnd
                if t >= NG:
                    continue
                cn, n0 = images_of(t), group_of(t) * Gi
                pr = idx - nF
                if pr >= cn * nsplit:
                    continue
                i, sp = divmod(pr, nsplit)
                yield None                                 # (the streaming part of the task)
                if t >= R:
                    while cntB[t - R] < images_of(t - R) * NS:
                        yield ("B", t - R, tk)
                ring[(t % R, i * nsplit + sp)] = t
                key = (n0 + i, sp)
                assert key not in reduced
                reduced.add(key)
                cntA[t] += 1
                yield None
            else:
                t = rnd - L
                if t < 0 or t >= NG:
                    continue
                cn, n0 = images_of(t), group_of(t) * Gi
                pr = idx - nF - nS
                if pr >= cn * nsplit:
                    continue
                while cntB[t] < cn * NS:
                    yield ("B", t, tk)
                i, sp = divmod(pr, nsplit)
                assert all((n0 + i, sl) in coef for sl in range(NS))
                key = (n0 + i, sp)
                assert key in reduced and key not in applied
                applied.add(key)
                yield None

    gens = {w: worker() for w in range(resident)}
    blocked = {}
    steps = 0
    while gens:
        def ready(w):
            if w not in blocked:
                return True
            kind, t, _ = blocked[w]
            return (cntA[t] >= images_of(t) * nsplit) if kind == "A" else (cntB[t] >= images_of(t) * NS)
        runnable = [w for w in gens if ready(w)]
        assert runnable, f"deadlock with {resident} resident work-groups: blocked on {sorted(set(blocked.values()))[:6]}"
        w = rng.choice(runnable)
        blocked.pop(w, None)
        try:
            r = next(gens[w])
            if r is not None:
                # a blocked task only ever waits for tasks with smaller tickets
                kind, t, tk = r
                blocked[w] = r
        except StopIteration:
            del gens[w]
        steps += 1
        assert steps < 20_000_000
    assert reduced == applied and len(coef) == n * NS and sorted(tails) == list(range(n_tail))
    assert len(reduced) == n * nsplit                       # every (image, row range) once


@pytest.mark.parametrize("shape", [(6, 32 * 32, 64), (9, 16 * 16, 128), (40, 24 * 24, 32), (12, 64 * 64, 128), (30, 32 * 32, 256)])
@pytest.mark.parametrize("lead", [1, 2, 4])
def test_queue_is_deadlock_free_and_ring_safe(shape, lead):
    """a small device (8 CUs: few tasks per round, several groups, the ring wraps many times); 1, 3 and 16 resident work-groups"""
    n, hw, c = shape
    ok, p = _plan(n, hw, c, cus=8)
    assert ok and p["NG"] >= 2
    for resident, seed in ((1, 0), (3, 1), (16, 2)):
        _simulate(n, hw, c, p, resident, seed, lead)


def test_queue_at_the_benched_plan_with_few_resident_work_groups():
    """the plan of 8 x 128 ch x 256^2 on 256 CUs (512 pairs per image) drained by 5 work-groups"""
    ok, p = _plan(8, 65536, 128)
    assert ok and p["nsplit"] == 512
    _simulate(8, 65536, 128, p, resident=5, seed=3)
