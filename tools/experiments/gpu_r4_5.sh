#!/bin/bash
# round 4, session 5: tiled weight pack (parity vs the gather kernel, step A/B), clock / power under load, r4 parity tests, 8-rank dry run
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=make-a-scene_amd
O=gpurun_out/r4_5; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_pack_tiles.py tests/test_gpu_parity_r4.py tests/test_gpu_parity_r2.py -x -q > $O/pytest_a.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_a.txt
grep -v "^  warn\|Warning\|amdgpu.ids\|^  bf16\|^  fp32" $O/pytest_a.txt | tail -30 | cut -c1-220
timeout 300 python bench.py --steps 20 --warmup 12 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('bench:', d['value'], 'img/s', d['ms_per_step'], 'ms/step; final_loss', d['final_loss'], 'peak GiB', d['peak_memory_gib'])" | tee $O/bench.txt
bash tools/probes/build.sh > /dev/null 2>&1
timeout 600 python tools/probes/clock_power.py > $O/clock_power.txt 2>&1; cat $O/clock_power.txt | cut -c1-220
ls /sys/class/drm/ 2>/dev/null | head; ls /sys/class/drm/card*/device/hwmon/hwmon*/ 2>/dev/null | head -40
for k in "conv_fwd --c 128 --hw 256" "wgrad --c 128 --hw 256"; do bash tools/pmc_kernel.sh "GRBM_GUI_ACTIVE" $k 2>&1 | tail -4; done > $O/gui_active.txt 2>&1; cat $O/gui_active.txt | cut -c1-250
timeout 1300 python -m pytest tests/test_gpu_dp.py -x -q -k "eight_ranks" > $O/pytest_8rank.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_8rank.txt; tail -15 $O/pytest_8rank.txt | cut -c1-250
