cd $GRAFT_REPO_ROOT
bash tools/prof_bench.sh r3 --steps 10 --warmup 2 > gpurun_out/prof_r3.log 2>&1
tail -30 gpurun_out/prof_r3.log
bash tools/prof_match.sh r3c4 match 11000000 6 8 2 2>&1 | tail -30
