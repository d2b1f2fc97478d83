# SQ wave-cycle split and LDS counters of the MFMA kernels of a step (WL, default full); summarised by scripts/summarize_sq.py into
# profiles/<tag>_pmc_sq.csv / _pmc_lds.csv.  Three separate --pmc passes (8 SQ slots per pass), kernel trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/sq_${RT:-r05}
WL=${WL:-full}
rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/a -o s -- python3 $R/bench.py --steps 4 --warmup 2 --workload $WL --no-cpu-baseline --no-kernel-pass > /dev/null 2> $O/a.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $O/b -o s -- python3 $R/bench.py --steps 4 --warmup 2 --workload $WL --no-cpu-baseline --no-kernel-pass > /dev/null 2> $O/b.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/c -o s -- python3 $R/bench.py --steps 4 --warmup 2 --workload $WL --no-cpu-baseline --no-kernel-pass > /dev/null 2> $O/c.err
ls $O/*; tail -2 $O/a.err
