# rocprofv3 summaries of the round-2 build (copied into profiles/ by hand afterwards)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r02
mkdir -p $O
cd $R
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r02 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o r02 -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-pass > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o r02 -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-pass > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -o r02 -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-pass > /dev/null 2> $O/pmc_mfma.err
find $O -name "*.csv" | head -30
du -sh $O
