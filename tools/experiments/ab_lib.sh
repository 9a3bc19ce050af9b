for rep in 1 2 3; do
for lib in ${LIBS:-libprt_hip_prev.so libprt_hip.so}; do
PRT_LIB=$PWD/pyrayt_amd/csrc/$lib python bench.py --no-cpu-baseline --side-steps 0 ${BENCH_ARGS:---steps 200 --warmup 20} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', 'value %.4e'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'launch us %.2f'%(d['roofline']['avg_launch_ms']*1e3), 'one_stream launch us %.2f'%((d['roofline'].get('one_stream') or {}).get('avg_launch_ms',0)*1e3), d['verified'])
"
done; done
