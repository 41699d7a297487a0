#!/usr/bin/env python3
"""Per-step HBM-side traffic from the per-dispatch counter files tools/step_traffic.sh writes.

    python tools/step_traffic.py <dir with pass_*.csv> K W [--json out.json]

The K timed steps are cut out of the dispatch stream by the one kernel that runs exactly once per step (`adam_multi_kernel`): the window
is (dispatch of Adam #W, dispatch of Adam #(W+K)].  Per kernel family: launches per step, FETCH_SIZE / WRITE_SIZE per step, and the read
bytes after the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts REQUESTS at 64 B each, so
a stream of full 128-byte lines reports half its bytes (x2: calibrated here on `gn_act`, whose read is exactly its tensor -- profiles/
r05_gn_tuning.txt -- and on tools/probes/fetch_calib.hip, profiles/r02_fetch_calibration.txt) while a stream of 64-byte requests (the
32-channel-chunk LDS-DMA pieces of the wide / sub-pixel / LDS-DMA weight-gradient kernels: 4 lanes x 16 B per pixel) reports all of
them (x1: same calibration file, profiles/r05_up2_pmc.txt).  Families without a calibration of their own are priced at x2 and flagged;
the totals are also given with x1 and x2 everywhere (lower / upper bound).  WRITE_SIZE is taken as is (equals the output tensor of the
convolution and of gn_act exactly).  Infinity-Cache hits are INCLUDED in both counters (they sit at the L2's fabric side), so this is
"bytes that left the XCDs' L2s", the quantity SURVEY 8(d)'s algorithmic bytes are compared with."""
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
    total = tot["read"] + tot["write"]
    print(f"{'TOTAL per step':58s} {sum(launches.values()) / K:8.1f} {'':10s} {'':4s} {tot['read'] / 1e9:8.2f} {tot['write'] / 1e9:9.2f} {total / 1e9:9.2f}")
    print(f"bounds: every read at x1 {tot['lo'] / 1e9:.2f} GB, every read at x2 {tot['hi'] / 1e9:.2f} GB per step")
    print(f"algorithmic (SURVEY 8(d): 1.79 GB per image x {batch}): {alg / 1e9:.2f} GB per step -> traffic / algorithmic = {total / alg:.2f} "
          f"(bounds {tot['lo'] / alg:.2f} .. {tot['hi'] / alg:.2f})")
    gn = sum(r["read_bytes"] + r["write_bytes"] for r in rows if r["family"].startswith("GroupNorm"))
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
            dom = dict(launches=len(sel) / K, fetch_size_bytes=fe, write_bytes=wr, hbm_bytes_per_launch=fe + wr, algorithmic_bytes=2 * 32 * 256 * 256 * 128 * 2)
            print(f"dominant launch in the step (wide kernel, 128->128 @256^2, B=32): {len(sel) / K:.0f} launches per step, FETCH_SIZE {fe / 1e6:.1f} MB (x1: 64-byte "
                  f"requests) + WRITE_SIZE {wr / 1e6:.1f} MB = {(fe + wr) / 1e6:.1f} MB per launch; algorithmic 1073.7 MB (residual launches +536.9 MB)")
    else:
        print("(dispatch sequences differ between passes: no per-launch figure)")
    if out_json:
        json.dump(dict(steps=K, warmup=W, per_gpu_batch=batch, step_traffic_bytes=total, read_bytes=tot["read"], write_bytes=tot["write"],
                       lower_bound_bytes=tot["lo"], upper_bound_bytes=tot["hi"], algorithmic_bytes=alg,
                       step_traffic_over_algorithmic=total / alg, groupnorm_bytes=gn, families=rows, read_bytes_by_request_size=exact, dominant_launch=dom,
                       source="tools/step_traffic.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (own passes) over bench.py's timed steps"),
                  open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
