"""Builds libmas_hip.so (gfx950) in-tree with hipcc.  `python -m mas_hip.build` or
`__graft_entry__.build()`.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libmas_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# per-file additions.  attention.hip: one fp32 VALU instruction per score in the softmax (no SLP packing into v_pk_*_f32: packed fp32
# VALU beside MFMAs is slower on gfx950 -- measured -2 ... -5 % on the forward kernel, profiles/r03_attn_v2.txt)
# conv3x3_wide.hip: the same for the epilogue's bias / residual adds (-2 % on the launch, gpu_r3_17.sh; the stream kernel measured 5 % SLOWER
# without packing and keeps it)
PER_FILE_FLAGS = {"attention.hip": ["-DFA_SCALAR_SOFTMAX", "-fno-slp-vectorize"], "conv3x3_wide.hip": ["-fno-slp-vectorize"],
                  "conv_up2.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "mas_hip.h"))
    return hdrs


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(p) <= t for p in sources() + _deps())


def build(force=False, verbose=True):
    if not force and up_to_date():
        return LIB
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    hdr_t = max(os.path.getmtime(p) for p in _deps())
    procs, objs = [], []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_t):
            continue
        cmd = [hipcc] + FLAGS + PER_FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print("[mas_hip.build]", " ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = []
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed.append((src, out))
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc failed:\n" + "\n".join(f"--- {s}\n{o}" for s, o in failed))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print("[mas_hip.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
