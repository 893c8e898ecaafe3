#!/usr/bin/env bash
# Turnstile path: parity cases, then bench lines with / without value forwarding (results are identical; timing switch),
# then -- if a -DQE_TURN_CLOCKS build of the engine lies at tools/libqe_turn_clocks.so -- where a launch spends its time.
# That build (in the container, before gpurun):  bash dist_classicrl_amd/csrc/build.sh EXTRA=-DQE_TURN_CLOCKS OBJ=build_clocks LIB=../../tools/libqe_turn_clocks.so
# (the later -o wins; *.so files are git-ignored but travel to the GPU box).
set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "turnstile" > gpurun_out/fwd_parity.log 2>&1 || { tail -30 gpurun_out/fwd_parity.log; exit 1; }
tail -2 gpurun_out/fwd_parity.log
for wl in c3 c4shard c5; do
  for f in 1 0 1; do
    timeout -k 10 120 python bench.py --workload $wl --steps 2000 --warmup 500 --no-cpu-baseline --turn-forward $f > gpurun_out/fwd_${wl}_$f.json 2>gpurun_out/fwd_${wl}_$f.err
    python - <<PY
import json
d=json.load(open("gpurun_out/fwd_${wl}_$f.json"))
print("$wl fwd=$f", round(d["value"]/1e6,1), "M env-steps/s; device region", round(d["device_region_ms"]/d["steps"]*1e3,2), "us/step; sampled launches", round(d["roofline"]["avg_launch_us"],2), "us")
PY
  done
done
if [ -f tools/libqe_turn_clocks.so ]; then
  for wl in c3 c5; do
    echo "== $wl, diagnostic build"
    QE_LIB_PATH=$PWD/tools/libqe_turn_clocks.so QE_PRINT_TURN_CLOCKS=1 timeout -k 10 120 python bench.py --workload $wl --steps 500 --warmup 2000 --no-cpu-baseline 2>&1 >/dev/null | grep "turn clocks" | tail -6 | tee gpurun_out/turn_clocks_$wl.txt
  done
fi
