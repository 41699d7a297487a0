#!/usr/bin/env python3
"""Static model of the VMEM counter around the barriers of the LDS-DMA kernels, from hipcc's assembly output (round 6, after the
conv_fwd.hip find: a compiler-made `__syncthreads()` published a weight stage with `s_waitcnt lgkmcnt(0); s_barrier` because the only
vmcnt(0) in front of it sat in an exec-masked region that one wave of the tile skips; profiles/r06_determinism.txt).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -o conv_fwd.s make-a-scene_amd/csrc/conv_fwd.hip
    python tools/check_lds_dma_waits.py conv_fwd.s [more.s ...]

Per kernel that issues `buffer_load ... lds`: the instruction stream is walked in text order (loop bodies a second time, entered with the
queue the back edge leaves), every VMEM instruction is queued, every `s_waitcnt vmcnt(N)` retires all but the newest N -- EXCEPT waits
inside an exec-masked region (`s_and_saveexec ... s_cbranch_execz L` up to `L:`), which a wave may skip and which therefore retire
nothing here.  For every `s_barrier` it prints how many LDS-DMAs are still queued.  A DMA in flight across a barrier is legitimate only
where the kernel's design says so (the wide / stream kernels keep the NEXT chunk's patch DMAs in flight; their weight DMAs must be
retired): compare with the comments at the kernel's wait macros.  `--max N`: exit 1 if any barrier of any kernel has more than N."""
import re
import sys


def kernels(path):
    cur, out = None, {}
    for line in open(path):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            cur = m.group(1)
            out[cur] = []
        elif line.startswith(".Lfunc_end"):
            cur = None
        elif cur is not None:
            out[cur].append(line.strip())
    return out


def walk(lines, lo, hi, queue, report, labels):
    skip_end = -1                                        # text position up to which the current instructions may be jumped over
    backedges = {}
    for i in range(lo, hi):
        t = lines[i]
        cond = i < skip_end
        if t.startswith("s_cbranch") or t.startswith("s_branch"):
            tgt = t.split()[1]
            if tgt in labels and labels[tgt] <= i:
                backedges[tgt] = (i, list(queue))
            elif tgt in labels and t.startswith("s_cbranch") and not any(x.startswith("s_barrier") for x in lines[i:labels[tgt]]):
                skip_end = max(skip_end, labels[tgt])    # an if-block (no barrier inside): what follows up to the label may not execute
        elif re.match(r"(buffer|global|flat|scratch)_(load|store|atomic)", t):
            if t.startswith("buffer_load") and " lds" in t:
                queue.append("D")
            elif not cond:
                queue.append("V")                        # (a skippable ordinary access must not make a later counted wait look deeper)
        elif t.startswith("s_waitcnt") and "vmcnt" in t and not cond:
            n = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
            del queue[:max(0, len(queue) - n)]
        elif t.startswith("s_barrier"):
            report[i] = max(report.get(i, 0), queue.count("D"))
    return backedges


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    limit = int(sys.argv[sys.argv.index("--max") + 1]) if "--max" in sys.argv else None
    args = [a for a in args if not a.isdigit()]
    worst = 0
    for path in args:
        for name, lines in kernels(path).items():
            if not any(t.startswith("buffer_load") and " lds" in t for t in lines):
                continue
            labels = {m.group(1): i for i, t in enumerate(lines) for m in [re.match(r"^(\.LBB\S+):", t)] if m}
            report = {}
            back = walk(lines, 0, len(lines), [], report, labels)
            for tgt, (pos, q) in back.items():              # loop bodies again, entered with what the back edge carries
                walk(lines, labels[tgt], pos + 1, q, report, labels)
            counts = [report[k] for k in sorted(report)]
            worst = max([worst] + counts)
            print(f"{path}: {name[:100]}\n    LDS-DMAs still queued at each s_barrier (text order): {counts}")
    return 1 if limit is not None and worst > limit else 0


if __name__ == "__main__":
    sys.exit(main())
