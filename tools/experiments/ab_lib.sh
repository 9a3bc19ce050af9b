for rep in 1 2 3; do
for lib in libprt_hip.so libprt_hip_ssc.so; do
PRT_LIB=$PWD/pyrayt_amd/csrc/$lib python bench.py --no-cpu-baseline --side-steps 0 --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', 'value %.4e'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'one_stream launch us %.2f'%(d['roofline']['one_stream']['avg_launch_ms']*1e3), 'value_one_stream %.4e'%d['value_one_stream'], d['verified'])
"
done; done
