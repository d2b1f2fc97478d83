# one fresh-lease sample of the driver's command without the conditioning phase (VERDICT r4 #1a): gpurun -- bash scripts/stall_lease.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-a}; O=$R/gpurun_out/lease_$T; mkdir -p $O; cd $R
EONERF_BENCH_CONDITION=0 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/first.json 2> $O/first.err
grep "^\[bench\]" $O/first.err | sed "s/^/$T first /"
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/cond.json 2> $O/cond.err
grep "^\[bench\]" $O/cond.err | sed "s/^/$T cond /"
