cd $GRAFT_REPO_ROOT
echo "# N = 1, 30,000 steps of the launcher (full EO-NeRF state: shadow pass, uncertainty loss from epoch 2, StepLR, epoch shuffles), device status checked every 5,000 steps"
timeout -k 10 400 python -m eonerf_code_amd.train_dp --synthetic_rays 1048576 --batch_size 4096 --n_images 19 --max_train_steps 30000 --check_every 5000 --logs_dir /tmp/soak1 --exp_name s1 2>&1 | grep -E "epoch=|Error|error|Traceback" | tail -9
echo
echo "# the same under an RCCL process group of world size 1 with the exchange forced (EONERF_FORCE_ALLREDUCE=1: two buckets, next step's sampler under the exchange), 30,000 steps"
MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 EONERF_FORCE_ALLREDUCE=1 timeout -k 10 400 python -m eonerf_code_amd.train_dp --synthetic_rays 1048576 --batch_size 4096 --n_images 19 --max_train_steps 30000 --check_every 5000 --logs_dir /tmp/soak2 --exp_name s2 2>&1 | grep -E "epoch=|Error|error|Traceback" | tail -9
