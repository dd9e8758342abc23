cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
(timeout 300 python tools/run_event_check.py small > gpurun_out/r3/small.log 2>&1; echo rc=$? >> gpurun_out/r3/small.log)
tail -4 gpurun_out/r3/small.log
(timeout 900 python tools/run_match_check.py scale 11000000 6 8 2 gpurun_out/r3/scale_c4.json > gpurun_out/r3/scale_c4.log 2>&1; echo rc=$? >> gpurun_out/r3/scale_c4.log)
cat gpurun_out/r3/scale_c4.log
(timeout 600 python tools/run_match_check.py scale 1650000 6 2 3 gpurun_out/r3/scale_c3.json > gpurun_out/r3/scale_c3.log 2>&1; echo rc=$? >> gpurun_out/r3/scale_c3.log)
cat gpurun_out/r3/scale_c3.log
(timeout 1500 python tools/run_match_check.py fidelity blobs6 60000 40 gpurun_out/r3/fid_blobs6.json > gpurun_out/r3/fid_blobs6.log 2>&1; echo rc=$? >> gpurun_out/r3/fid_blobs6.log)
cat gpurun_out/r3/fid_blobs6.log
