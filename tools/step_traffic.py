#!/usr/bin/env python3
"""Per-step fabric-side traffic (L2 <-> Infinity Cache / HBM) from the per-dispatch counter files tools/step_traffic.sh writes
(VERDICT r5 next #2).

    python tools/step_traffic.py <dir with pass_*.csv> K W [--json out.json]

The K timed steps are cut out of the dispatch stream by the one kernel that runs exactly once per step (`adam_multi_kernel`): the window
is (dispatch of Adam #W, dispatch of Adam #(W+K)].  Per kernel family: launches per step, read and write bytes per step.

READ bytes.  FETCH_SIZE counts the L2's fabric-side read REQUESTS at 64 B each (/opt/skills/guides/MI355X_MICROARCH.md, HBM section:
"double it"), and this rocprofv3 also exposes the request-size counters: over whole steps TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ (32-B: 0,
64-B: < 0.1 %) for EVERY family -- the L2 fills whole 128-byte lines whatever part of the line was asked for.  So read bytes =
128 x RDREQ = 2 x FETCH_SIZE everywhere.  Rounds 2-5 priced the wide / sub-pixel / LDS-DMA weight-gradient kernels at x1 because
tools/probes/fetch_calib.hip's half-line stream (64 B of every 128-B line) reports exactly the bytes it ASKED for; what that calibration
shows is one 128-byte fill per line touched, i.e. twice the useful bytes crossed the fabric (their 32-channel-chunk patch pieces are
64 B per pixel: every line is touched by two chunk passes ~8 us apart, and the second touch misses the XCD's 4 MiB L2 about one time in
three -- 32 CUs stream ~3.5 MB through it in that time).  The x1 column is kept for comparison with the older profiles.
WRITE bytes: WRITE_SIZE as is (all write requests are 64 B: TCC_EA0_WRREQ_64B == TCC_EA0_WRREQ; equals the output tensors exactly).
Infinity-Cache hits are INCLUDED in both (the counters sit at the L2's fabric side): this is "bytes that left the XCDs' L2s", an upper
bound of the HBM bytes and the quantity SURVEY 8(d)'s algorithmic bytes are compared with."""
import collections
import csv
import json
import os
import sys

ALGORITHMIC_BYTES_PER_IMAGE = 1.79e9          # VQ-IMG 256^2 fwd + bwd, bf16

# (substring of the kernel name, family, FETCH_SIZE factor, calibrated?)
FAMILIES = [
    ("conv3x3_wide", "conv 3x3 s1 wide (fwd + dgrad)", 1, True),
    ("conv_up2", "Upsample conv, sub-pixel (fwd + dgrad)", 1, True),
    ("conv_wgrad_dma", "weight gradient, LDS-DMA", 1, True),
    ("wgrad_reduce", "split-K reduce", 2, False),
    ("gn_bwd", "GroupNorm backward passes", 2, True),
    ("gn_act", "GroupNorm+SiLU activation pass", 2, True),
    ("gn_", "GroupNorm statistics / finalize / small maps", 2, True),
    ("conv3x3_stream", "conv 3x3 s1 stream (16x16 maps)", 2, False),
    ("conv_s2", "Downsample conv (fwd, dgrad, wgrad)", 2, False),
    ("wgrad_s2", "Downsample conv (fwd, dgrad, wgrad)", 2, False),
    ("conv_thin", "conv_in / conv_out", 2, False),
    ("wgrad_thin", "conv_in / conv_out", 2, False),
    ("conv1x1", "1x1 convolutions", 2, False),
    ("wgrad1x1", "1x1 convolutions", 2, False),
    ("conv_fwd_kernel", "general conv kernels", 2, False),
    ("conv_wgrad", "general conv kernels", 2, False),
    ("spatial_attn", "spatial attention", 2, False),
    ("vq_", "vector quantiser", 2, False),
    ("bn_", "BatchNorm", 2, False),
    ("pack_weight", "weight image pack", 2, False),
    ("adam", "Adam", 2, False),
]


def family(name):
    for sub, fam, fac, cal in FAMILIES:
        if sub in name:
            return fam, fac, cal
    return "other (ATen glue: casts, pads, copies, fills, L1 loss)", 2, False


def load(path):
    """-> ordered list of (dispatch id, kernel name, {counter: value})"""
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        k = int(r["Dispatch_Id"])
        ent = d.setdefault(k, [r["Kernel_Name"], collections.defaultdict(float)])
        ent[1][r["Counter_Name"]] += float(r["Counter_Value"])
    return [(k, v[0], v[1]) for k, v in sorted(d.items())]


def window(disp, K, W):
    adam = [i for i, (_, n, _) in enumerate(disp) if "adam_multi" in n]
    if len(adam) < W + K:
        raise SystemExit(f"only {len(adam)} Adam dispatches in the run, need {W + K}")
    lo = adam[W - 1] + 1 if W > 0 else 0
    hi = adam[W + K - 1] + 1
    return disp[lo:hi]


def main():
    d, K, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    meta = {}
    counters_seen = set()
    windows = []
    for f in sorted(os.listdir(d)):
        if not (f.startswith("pass_") and f.endswith(".csv")):
            continue
        disp = window(load(os.path.join(d, f)), K, W)
        windows.append(disp)
        first = not launches
        for _, name, c in disp:
            fam, fac, cal = family(name)
            meta[fam] = (fac, cal)
            if first:
                launches[fam] += 1
            for cn, v in c.items():
                per[fam][cn] += v
                counters_seen.add(cn)
    if "FETCH_SIZE" not in counters_seen or "WRITE_SIZE" not in counters_seen:
        raise SystemExit(f"need FETCH_SIZE and WRITE_SIZE passes, have {sorted(counters_seen)}")
    batch = 32
    alg = ALGORITHMIC_BYTES_PER_IMAGE * batch
    print(f"# {K} timed steps (after {W} warm-up steps), per STEP; FETCH_SIZE / WRITE_SIZE in KB as reported; read = FETCH_SIZE x 1024 x factor")
    print(f"{'family':58s} {'launches':>8s} {'FETCH MB':>10s} {'fac':>4s} {'read GB':>8s} {'write GB':>9s} {'total GB':>9s}  note")
    tot = dict(read=0.0, write=0.0, lo=0.0, hi=0.0)
    rows = []
    for fam in sorted(per, key=lambda k: -(per[k]["FETCH_SIZE"] * meta[k][0] + per[k]["WRITE_SIZE"])):
        fac, cal = meta[fam]
        fetch = per[fam]["FETCH_SIZE"] * 1024 / K
        write = per[fam]["WRITE_SIZE"] * 1024 / K
        read = fetch * fac
        tot["read"] += read; tot["write"] += write; tot["lo"] += fetch + write; tot["hi"] += 2 * fetch + write
        rows.append(dict(family=fam, launches_per_step=launches[fam] / K, fetch_size_bytes=fetch, factor=fac, calibrated=cal, read_bytes=read,
                         write_bytes=write))
        print(f"{fam:58s} {launches[fam] / K:8.1f} {fetch / 1e6:10.1f} {fac:4d} {read / 1e9:8.2f} {write / 1e9:9.2f} {(read + write) / 1e9:9.2f}"
              f"  {'' if cal else 'factor uncalibrated'}")
    lines128 = {"TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_128B_sum"} <= counters_seen and \
        abs(sum(per[f]["TCC_EA0_RDREQ_128B_sum"] for f in per) / max(sum(per[f]["TCC_EA0_RDREQ_sum"] for f in per), 1.0) - 1.0) < 0.01
    for r in rows:                       # every fabric read request is a 128-byte line fill (checked above): read = 2 x FETCH_SIZE
        r["read_bytes_x1_model"] = r["read_bytes"]
        r["read_bytes"] = 2 * r["fetch_size_bytes"]
    total = tot["hi"]
    tot["read_model"], tot["read"] = tot["read"], tot["hi"] - tot["write"]
    print(f"{'TOTAL per step (old per-family factors)':58s} {sum(launches.values()) / K:8.1f} {'':10s} {'':4s} {tot['read_model'] / 1e9:8.2f} {tot['write'] / 1e9:9.2f} "
          f"{(tot['read_model'] + tot['write']) / 1e9:9.2f}")
    print(f"TOTAL per step, every read request a 128-byte line fill ({'confirmed by TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ' if lines128 else 'request-size pass missing: the guide rule'}): "
          f"read {tot['read'] / 1e9:.2f} GB + write {tot['write'] / 1e9:.2f} GB = {total / 1e9:.2f} GB")
    print(f"algorithmic (SURVEY 8(d): 1.79 GB per image x {batch}): {alg / 1e9:.2f} GB per step -> traffic / algorithmic = {total / alg:.2f} "
          f"(with the old x1 factors: {(tot['read_model'] + tot['write']) / alg:.2f})")
    gn = sum(r["read_bytes"] + r["write_bytes"] for r in rows if r["family"].startswith("GroupNorm"))
    conv = sum(r["read_bytes"] + r["write_bytes"] for r in rows if "conv" in r["family"] or "gradient" in r["family"] or "reduce" in r["family"])
    print(f"convolution kernels (all families, split-K reduce included): {conv / 1e9:.2f} GB per step = {conv / alg:.2f} x the step's algorithmic bytes")
    print(f"GroupNorm passes: {gn / 1e9:.2f} GB per step = {gn / total:.2f} of the step's traffic (algorithmic bytes by SURVEY's rule: 0)")
    for extra in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_EA0_WRREQ_sum",
                  "TCC_EA0_WRREQ_64B_sum"):
        if extra in counters_seen:
            print(f"{extra}: {sum(per[f][extra] for f in per) / K:.4g} per step")
    exact = None
    if {"TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"} <= counters_seen:
        # request-size counters: the cross-check of the per-family factors (and the exact figure if the three sizes partition RDREQ)
        print("# read requests by size, per step and family: n(all) n32 n64 n128 | 32 n32 + 64 n64 + 128 n128 (GB) | same with n64 := all - n32 - n128 (GB)"
              " | factor-model read (GB)")
        exact = 0.0
        for r in rows:
            c = per[r["family"]]
            n, n32, n64, n128 = (c[k] / K for k in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"))
            b1 = 32 * n32 + 64 * n64 + 128 * n128
            b2 = 32 * n32 + 64 * max(n - n32 - n128, 0.0) + 128 * n128
            r["read_bytes_by_request_size"] = b2
            exact += b2
            print(f"{r['family']:58s} {n:10.4g} {n32:10.4g} {n64:10.4g} {n128:10.4g} | {b1 / 1e9:8.2f} | {b2 / 1e9:8.2f} | {r['read_bytes'] / 1e9:8.2f}")
        print(f"read bytes by request size: {exact / 1e9:.2f} GB per step (factor model: {tot['read'] / 1e9:.2f}); with the writes: {(exact + tot['write']) / 1e9:.2f} GB"
              f" = {(exact + tot['write']) / alg:.2f} x algorithmic")
    # the dominant launch IN the step (3x3 s1 128->128 @256^2, B = 32: 536.9 MB in, 536.9 MB out): the wide-kernel dispatches whose WRITE_SIZE
    # is that output (+ the 4 MB statistics table of the forward variants), matched across passes by position in the step window
    dom = None
    names = [[n for _, n, _ in w] for w in windows]
    if all(nm == names[0] for nm in names):
        merged = [collections.defaultdict(float) for _ in names[0]]
        for w in windows:
            for i, (_, _, c) in enumerate(w):
                for cn, v in c.items():
                    merged[i][cn] += v
        sel = [m for n, m in zip(names[0], merged) if "conv3x3_wide" in n and 520000 <= m["WRITE_SIZE"] <= 545000]
        if sel:
            fe = sum(m["FETCH_SIZE"] for m in sel) / len(sel) * 1024
            wr = sum(m["WRITE_SIZE"] for m in sel) / len(sel) * 1024
            n_res = sum(1 for m in sel if m["WRITE_SIZE"] > 535000)          # (the residual variants also write their statistics table)
            alg_l = (2 * 32 * 256 * 256 * 128 * 2 * len(sel) + 32 * 256 * 256 * 128 * 2 * n_res) / len(sel)
            dom = dict(launches=len(sel) / K, fetch_size_bytes=fe, read_bytes=2 * fe, write_bytes=wr, hbm_bytes_per_launch=2 * fe + wr,
                       algorithmic_bytes=alg_l, residual_launches=n_res / K)
            print(f"dominant launch in the step (wide kernel, 128->128 @256^2, B=32): {len(sel) / K:.0f} launches per step ({n_res / K:.0f} with a residual), "
                  f"FETCH_SIZE {fe / 1e6:.1f} MB -> read {2 * fe / 1e6:.1f} MB (128-byte line fills) + WRITE_SIZE {wr / 1e6:.1f} MB = {(2 * fe + wr) / 1e6:.1f} MB "
                  f"per launch; algorithmic {alg_l / 1e6:.1f} MB (1073.7, +536.9 for a residual) -> {(2 * fe + wr) / alg_l:.2f} x")
    else:
        print("(dispatch sequences differ between passes: no per-launch figure)")
    if out_json:
        json.dump(dict(steps=K, warmup=W, per_gpu_batch=batch, step_traffic_bytes=total, read_bytes=tot["read"], write_bytes=tot["write"],
                       all_read_requests_are_128B_lines=bool(lines128), with_old_x1_factors_bytes=tot["read_model"] + tot["write"], algorithmic_bytes=alg,
                       step_traffic_over_algorithmic=total / alg, groupnorm_bytes=gn, families=rows, read_bytes_by_request_size=exact, dominant_launch=dom,
                       source="tools/step_traffic.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (own passes) over bench.py's timed steps"),
                  open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
