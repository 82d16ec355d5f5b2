bash scripts/collect_profiles.sh r03 2>&1 | tail -8
ls gpurun_out/r03_summary
