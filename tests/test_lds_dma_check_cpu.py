"""tools/check_lds_dma_waits.py on two synthetic instruction streams: the barrier hole conv_fwd.hip had until round 6 (the only vmcnt(0) in
front of the barrier sits behind a forward `s_cbranch_execz`, so a wave may reach the barrier with its LDS-DMAs in flight) must be reported,
its fixed form (an unconditional `s_waitcnt vmcnt(0) lgkmcnt(0)`) and a designed counted wait must not."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("check_lds_dma_waits", os.path.join(ROOT, "tools", "check_lds_dma_waits.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


HOLE = """
_Z6kernelv:
	s_barrier
	global_load_dwordx4 v[2:5], v[6:7], off
	buffer_load_dwordx4 v1, s[0:3], s4 offen lds
	buffer_load_dwordx4 v1, s[0:3], s5 offen lds
.LBB0_1:
	s_waitcnt vmcnt(1)
	s_and_saveexec_b64 s[8:9], s[10:11]
	s_cbranch_execz .LBB0_3
	s_waitcnt vmcnt(0)
	ds_write_b128 v8, v[2:5]
.LBB0_3:
	s_or_b64 exec, exec, s[8:9]
	s_waitcnt lgkmcnt(0)
	s_barrier
	global_load_dwordx4 v[2:5], v[6:7], off
	buffer_load_dwordx4 v1, s[0:3], s4 offen lds
	buffer_load_dwordx4 v1, s[0:3], s5 offen lds
	s_cbranch_vccz .LBB0_1
	s_endpgm
.Lfunc_end0:
"""
FIXED = HOLE.replace("\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n\tglobal_load", "\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\tglobal_load")
COUNTED = """
_Z6kernelv:
	s_barrier
.LBB0_1:
	buffer_load_dwordx4 v1, s[0:3], s4 offen lds
	buffer_load_dwordx4 v1, s[0:3], s5 offen lds
	global_load_dwordx4 v[2:5], v[6:7], off
	s_waitcnt vmcnt(1) lgkmcnt(0)
	s_barrier
	s_cbranch_vccz .LBB0_1
	s_endpgm
.Lfunc_end0:
"""


def _counts(text, tmp_path, name):
    tool = _tool()
    p = tmp_path / name
    p.write_text(text)
    (kname, lines), = tool.kernels(str(p)).items()
    labels = {}
    import re
    for i, t in enumerate(lines):
        m = re.match(r"^(\.LBB\S+):", t)
        if m:
            labels[m.group(1)] = i
    report = {}
    back = tool.walk(lines, 0, len(lines), [], report, labels)
    for tgt, (pos, q) in back.items():
        tool.walk(lines, labels[tgt], pos + 1, q, report, labels)
    return [report[k] for k in sorted(report)]


def test_barrier_behind_a_skippable_wait_is_reported(tmp_path):
    assert _counts(HOLE, tmp_path, "hole.s") == [0, 1]            # the loop's barrier: the newer weight DMA may still be in flight (vmcnt(1) is the last wait every wave executes)


def test_unconditional_wait_clears_it(tmp_path):
    assert _counts(FIXED, tmp_path, "fixed.s") == [0, 0]


def test_counted_wait_retires_the_dmas_older_than_the_allowance(tmp_path):
    assert _counts(COUNTED, tmp_path, "counted.s") == [0, 0]      # vmcnt(1): only the ordinary load issued AFTER the DMAs stays queued
