# the 41 ms the driver's BENCH_r04 lost in its first timed block.  Run as the FIRST GPU work of a fresh lease:
#   gpurun -- bash scripts/stall_hunt.sh <tag>
# 1) the driver's exact command without the conditioning phase (the reproduction), 2) the same again (fresh process, warm box),
# 3) after 30 s of idle, 4) beside a 1-Hz rocm-smi poll (the driver polls smi while the bench runs), 5) chain + GEMM backward
# (EONERF_PIPE=0: no persistent launch), 6) with the conditioning phase (the shipped default).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-a}; O=$R/gpurun_out/stall_$T; mkdir -p $O; cd $R
run() { # name, env...
  local name=$1; shift
  env "$@" python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass > $O/$name.json 2> $O/$name.err
  grep "^\[bench\]" $O/$name.err | sed "s/^/$name /"
}
run 1_first EONERF_BENCH_CONDITION=0
run 2_again EONERF_BENCH_CONDITION=0
sleep 30
run 3_after_idle EONERF_BENCH_CONDITION=0
( while true; do rocm-smi -a --json > $O/smi_last.json 2>/dev/null; sleep 1; done ) & SMI=$!
run 4_smi_poll EONERF_BENCH_CONDITION=0
kill $SMI; wait $SMI 2>/dev/null
run 5_no_pipe EONERF_BENCH_CONDITION=0 EONERF_PIPE=0
sleep 30
run 6_conditioned EONERF_BENCH_CONDITION=1
