#!/usr/bin/env python3
"""Instruction mix and s_waitcnt histogram per kernel of a hipcc -save-temps gfx950 .s file.
    python tools/isa_stats.py file.s [name-substring]"""
import re
import sys
from collections import Counter

KEYS = ["v_mfma_f32_32x32x16_bf16", "v_mfma_f32_16x16x32_bf16", "ds_read_b128", "ds_read_b64_tr_b16", "ds_write_b128", "ds_write_b64",
        "global_load_lds_dwordx4", "buffer_load_dwordx4", "buffer_store_dwordx4", "global_load_dwordx4", "global_load_dwordx2",
        "global_store_dwordx4", "global_atomic_add_f32", "s_barrier", "scratch_load_dword", "scratch_store_dword", "scratch_load_dwordx4",
        "scratch_store_dwordx4", "v_readlane_b32", "v_writelane_b32", "v_accvgpr_write_b32", "v_accvgpr_read_b32", "s_nop", "v_exp_f32"]


def main():
    s = open(sys.argv[1]).read()
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    parts = re.split(r"\n(?=_Z\w+:)", s)
    for f in parts[1:]:
        name = f.split(":")[0]
        if "kernel" not in name or sub not in name:
            continue
        end = f.find(".Lfunc_end")
        body = f[:end] if end > 0 else f
        ops = re.findall(r"^\s+([a-z_0-9]+)", body, re.M)
        c = Counter(ops)
        print(name[-70:], {k: c[k] for k in KEYS if c[k]}, "total", len(ops))
        print("   waits:", dict(Counter(re.findall(r"s_waitcnt ([^\n;]*)", body))))


if __name__ == "__main__":
    main()
