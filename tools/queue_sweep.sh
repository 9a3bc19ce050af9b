# EXPERIMENT (GPU box): traces in flight against the HIP runtime's number of hardware queues; results in profiles/r3/streams.txt
for rays in 125000 250000 500000 1000000; do
 for q in 4 8; do
  for st in 2 3 4; do
    GPU_MAX_HW_QUEUES=$q python bench.py --rays $rays --steps 300 --warmup 20 --no-cpu-baseline --side-steps 0 --streams $st 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rays $rays hw_queues $q streams $st: ms/step %.4f value %.3e'%(d['ms_per_step'], d['value']))"
  done
 done
done
