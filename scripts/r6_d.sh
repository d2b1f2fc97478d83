O=gpurun_out/r6d; mkdir -p $O
timeout -k 10 200 python scripts/dbg_ns.py > $O/dbg_ns.txt 2>&1; tail -8 $O/dbg_ns.txt
timeout -k 10 900 bash scripts/coresidency.sh > $O/coresidency.txt 2>&1; cat $O/coresidency.txt | tail -30
