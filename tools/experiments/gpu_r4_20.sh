cd "$GRAFT_REPO_ROOT"; export PYTHONPATH=make-a-scene_amd
for t in 1 0; do for r in 0 1; do
MAS_CONV_STREAM_TH8=$t python tools/kbench.py conv_fwd --n 32 --c 512 --hw 16 --res $r --iters 100 2>&1 | grep "^conv_fwd" | sed "s/^/th8=$t res=$r /"
done; done
