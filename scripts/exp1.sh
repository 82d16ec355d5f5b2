cd $GRAFT_REPO_ROOT
STEPS=100 SWEEP="fir_T=288,384,480,576,768,960;stage_mask=1" python scripts/pipe_experiment.py 2>&1 | grep -v amdgpu.ids
STEPS=100 SWEEP="fir_T=384,576,768;stage_mask=31" python scripts/pipe_experiment.py 2>&1 | grep -v amdgpu.ids
