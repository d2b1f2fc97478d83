# rocprofv3 summaries of the current build (gpurun_out/prof_$R3 -> copied into profiles/ by scripts/summarize_profiles.py; see profiles/README.md)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
RT=${RT:-r05}
O=$R/gpurun_out/prof_$RT
rm -rf $O; mkdir -p $O
cd $R
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
for WL in rgb full; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$WL -o prof -- python3 bench.py --steps 20 --warmup 5 --workload $WL --no-cpu-baseline --no-kernel-pass > $O/bench_under_rocprof_$WL.json 2> $O/stats_$WL.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$WL -o prof -- python3 bench.py --steps 4 --warmup 2 --workload $WL --no-cpu-baseline --no-kernel-pass > /dev/null 2> $O/pmc_fetch_$WL.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$WL -o prof -- python3 bench.py --steps 4 --warmup 2 --workload $WL --no-cpu-baseline --no-kernel-pass > /dev/null 2> $O/pmc_write_$WL.err
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma_$WL -o prof -- python3 bench.py --steps 4 --warmup 2 --workload $WL --no-cpu-baseline --no-kernel-pass > /dev/null 2> $O/pmc_mfma_$WL.err
done
EONERF_PIPE=0 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_chain_gemm_path.json 2> /dev/null
python3 bench.py --steps 10 --warmup 3 --precision fp32 --no-cpu-baseline > $O/bench_fp32.json 2> /dev/null
du -sh $O
