# round 6, first GPU call: the new tests, the twin-training experiment, a baseline bench of the round's first build
O=gpurun_out/r6a; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_n_samples.py::test_rendering_backward_runs_under_its_forwards_step_size tests/test_trainer_gpu.py -m gpu -q -x -s > $O/tests_new.log 2>&1; echo "rc=$?" >> $O/tests_new.log
tail -5 $O/tests_new.log
timeout -k 10 500 python scripts/twin_training.py 2000 > $O/twin.json 2> $O/twin.err; echo "twin rc=$?"; cat $O/twin.err | tail -6
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; grep '^\[bench\]' $O/bench.err | tail -4
