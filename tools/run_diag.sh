#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/diag
for knob in NONE PRT_NO_CHAIN; do echo "=== $knob"; env $knob=1 python tools/diag_fixture.py adv_lens adv_stop adv_prism adv_condenser stale_box 2>&1 | grep -v amdgpu.ids; done > gpurun_out/diag/diag.txt
cat gpurun_out/diag/diag.txt
