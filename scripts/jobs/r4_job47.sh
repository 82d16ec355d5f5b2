# round 4, job 47: the round's profiles on the final tree (K3 register request, formatter out of scratch, stage-mask legs in the line)
mkdir -p gpurun_out/r4
bash scripts/collect_profiles.sh r04 > gpurun_out/r4/job47_collect.log 2>&1
tail -2 gpurun_out/r4/job47_collect.log | cut -c1-300
