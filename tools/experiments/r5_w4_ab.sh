#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r5_gpu1; mkdir -p $out
cd $R
{ for w in early late after_available after_count; do python3 tools/queue_probe.py 1 $w 2>&1 | grep -v amdgpu.ids; done
  python3 tools/queue_probe.py 8 late 2>&1 | grep -v amdgpu.ids; python3 tools/queue_probe.py 2 early 2>&1 | grep -v amdgpu.ids; } > $out/queue_probe.txt 2>&1
python3 -m pytest tests/test_gpu_custom_materials.py tests/test_gpu_frame.py -m gpu -q -x > $out/new_tests.txt 2>&1
L=$R/pyrayt_amd/csrc
python3 tools/ab.py --reps 3 "exact::--streams 1 --side-steps 0" "w4:PRT_LIB=$L/libprt_hip_w4.so:--streams 1 --side-steps 0" \
  "fastw4:PRT_LIB=$L/libprt_hip_fastw4.so:--streams 1 --side-steps 0" "fast2w4:PRT_LIB=$L/libprt_hip_fast2w4.so:--streams 1 --side-steps 0" > $out/config2_w4.txt 2>&1
C3="--workload config3 --rays 4000000 --steps 50 --warmup 5 --side-steps 0"
python3 tools/ab.py --reps 3 "exact::$C3" "w4:PRT_LIB=$L/libprt_hip_w4.so:$C3" \
  "fastw4:PRT_LIB=$L/libprt_hip_fastw4.so:$C3" "fast2w4:PRT_LIB=$L/libprt_hip_fast2w4.so:$C3" > $out/config3_w4.txt 2>&1
python3 -m pytest tests -m gpu -q -x > $out/gpu_suite.txt 2>&1
tail -5 $out/*.txt
