#!/bin/bash
# One GPU-box pass that regenerates everything under profiles/<round>/ from the current build:
#   bench line, rocprofv3 kernel-trace stats of the same command, HBM traffic (PMC, calibrated),
#   SQ issue/stall counters of k_generation and k_hit, renderer kernel stats, the other BASELINE configs at
#   their per-GPU sizes (bench line + kernel stats each), the per-rank shard table of the 1/2/4/8-GPU curve,
#   the shortcut counters of the shipping library, the cull-step tables and the nearest-hit kernel A/B.
# usage (through gpurun): bash tools/refresh_profiles.sh ; then copy gpurun_out/refresh/* into profiles/<round>/
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/refresh
rm -rf $out $R/gpurun_out/traffic $R/gpurun_out/sq; mkdir -p $out
cd $R
(cd tools/ubench && make -s copy_f64 >/dev/null 2>&1)
python3 bench.py > $out/bench.json 2> $out/bench.err
# ... and as the round driver launches it
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_form.json 2>> $out/bench.err
cd /tmp
# the same command under rocprofv3: per-kernel stats, and from the per-dispatch timestamps the UNION of the overlapping
# k_generation dispatches (tools/busy_union.py) -- what roofline.kernel_ms_per_step / avg_launch_ms have to agree with
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o trace -- python3 $R/bench.py --no-cpu-baseline --side-steps 0 > $out/trace.log 2>&1
python3 $R/tools/busy_union.py $(find $out -name "trace_kernel_trace.csv" | head -1) > $out/busy_union.txt 2>&1
grep "^{" $out/trace.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench.py under rocprofv3 (this run): ms_per_step %.4f kernel_ms_per_step %.4f avg_launch_ms %.5f (= union / launches); its traces\' own intervals per launch %.5f; one stream %.5f' % (d['ms_per_step'], r['kernel_ms_per_step'], r['avg_launch_ms'], r['busy']['avg_launch_ms_on_its_stream'], r['one_stream']['avg_launch_ms']))" >> $out/busy_union.txt
# the same steps on ONE stream: the generation kernel's own duration (what roofline.avg_launch_ms is measured on)
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o trace_one_stream -- python3 $R/bench.py --no-cpu-baseline --side-steps 0 --streams 1 > $out/trace_one_stream.log 2>&1
for cfg in "config3 4000000" "config4 8000000" "config5 2000000"; do
  set -- $cfg
  python3 $R/bench.py --workload $1 --rays $2 --steps 50 --warmup 5 --no-cpu-baseline > $out/bench_$1.json 2>> $out/bench.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o trace_$1 -- python3 $R/bench.py --workload $1 --rays $2 --steps 50 --warmup 5 --no-cpu-baseline --side-steps 0 --streams 1 > $out/trace_$1.log 2>&1
done
# BASELINE config 4 WHOLE on one GPU: 8 wavelengths x 8M rays = 64M rays, 192M rows (generation_limit 4: the job has three generations)
python3 $R/bench.py --workload config4 --rays 64000000 --generation-limit 4 --steps 6 --warmup 2 --reps 3 --side-steps 2 --no-cpu-baseline > $out/bench_config4_full.json 2>> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o render -- python3 $R/tools/render_bench.py --width 640 8192 --cpu-pixels 20000 > $out/render.log 2>&1
cd $R
bash tools/traffic.sh > $out/traffic.log 2>&1
bash tools/sq.sh gen python3 $R/bench.py --steps 3 --warmup 1 --spinup-ms 0 --no-cpu-baseline --side-steps 0 --no-pipeline > /dev/null 2>&1
bash tools/sq.sh hit python3 $R/tools/hit_only.py 6 > /dev/null 2>&1
python3 tools/sq.py gpurun_out/sq gen > $out/sq_counters.txt
python3 tools/sq.py gpurun_out/sq hit >> $out/sq_counters.txt
python3 tools/traffic.py gpurun_out/traffic $out/traffic.json > /dev/null
cp gpurun_out/traffic/bench_FETCH_SIZE_counter_collection.csv $out/k_generation_pmc_FETCH_SIZE.csv
cp gpurun_out/traffic/bench_WRITE_SIZE_counter_collection.csv $out/k_generation_pmc_WRITE_SIZE.csv
cp gpurun_out/traffic/cal_FETCH_SIZE_counter_collection.csv $out/calibration_pmc_FETCH_SIZE.csv
cp gpurun_out/traffic/cal_WRITE_SIZE_counter_collection.csv $out/calibration_pmc_WRITE_SIZE.csv
# round 6: what a record plan buys (rows of the detector only / sums only) and the design loop it is for
python3 tools/plan_bench.py 2>&1 | grep -v amdgpu.ids > $out/plan_bench.txt
python3 tools/design_loop.py 2>&1 | grep -v amdgpu.ids > $out/design_loop.txt
python3 tools/shard_scaling.py > $out/shard_scaling.txt 2>&1
{ echo; python3 tools/c_host_bench.py 2>&1 | grep -v amdgpu.ids; } >> $out/shard_scaling.txt
python3 tools/slow_paths.py 2>&1 | grep -v amdgpu.ids > $out/slow_paths.txt
{ echo "# tools/cull_scaling.py (lens train listed along the axis)"; python3 tools/cull_scaling.py --counts 1 2 4 8 16 32 2>&1 | grep -v amdgpu.ids;
  echo; echo "# tools/cull_scaling.py --shuffle (lenses listed in random order, beams from both ends)"; python3 tools/cull_scaling.py --shuffle --counts 8 32 2>&1 | grep -v amdgpu.ids; } > $out/cull_scaling.txt
python3 tools/hit_ab.py 2>&1 | grep -v amdgpu.ids > $out/hit_variants.txt
{ python3 tools/aux_bench.py; python3 tools/frame_bench.py; python3 tools/gather_bench.py; } 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | grep -v amdgpu.ids > $out/aux_kernels.txt
# round 4: per-generation instruction counts / launch times, and SQ + PMC counters of the other BASELINE configs
bash tools/gen_counters.sh config2 1000000 > /dev/null 2>&1
bash tools/gen_counters.sh config3 4000000 > /dev/null 2>&1
cp gpurun_out/gen_counters/config2_per_generation.txt $out/per_generation_config2.txt
cp gpurun_out/gen_counters/config3_per_generation.txt $out/per_generation_config3.txt
bash tools/counters_cfg.sh > $out/counters_cfg.log 2>&1
cp gpurun_out/counters/sq_counters_config*.txt gpurun_out/counters/traffic_config*.json gpurun_out/counters/gen_times_config*.txt $out/
# per-dispatch durations with and without the sparse-loss forms (hint modes 4 / 5 / 6)
bash tools/gen_trace.sh config3 4000000 > /dev/null 2>&1
EXTRA="--flags 1024" LABEL=nokeep bash tools/gen_trace.sh config3 4000000 > /dev/null 2>&1
EXTRA="--ray-sets 4" LABEL=rotating bash tools/gen_trace.sh config2 1000000 > /dev/null 2>&1
EXTRA="--ray-sets 4 --flags 1024" LABEL=rotating_nokeep bash tools/gen_trace.sh config2 1000000 > /dev/null 2>&1
python3 tools/gen_durations.py gpurun_out/gen_trace > $out/gen_durations.txt
# keep the summaries, drop the bulky per-dispatch traces
rm -f $out/*_kernel_trace.csv $out/*_agent_info.csv $out/*domain_stats.csv
ls -la $out | head -60
tail -1 $out/bench.json | cut -c1-600
