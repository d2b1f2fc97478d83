O=gpurun_out/r6k; mkdir -p $O
timeout -k 10 500 bash scripts/pipe_stamps_abl.sh 128 8 136 > $O/stamps_abl.txt 2>&1; tail -60 $O/stamps_abl.txt
timeout -k 10 600 python scripts/twin_seeds.py 6 2000 > $O/twin_seeds.json 2> $O/twin_seeds.err; tail -3 $O/twin_seeds.err; tail -25 $O/twin_seeds.json
