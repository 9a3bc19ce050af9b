for rep in 1 2 3; do
for st in 2 3; do
python bench.py --no-cpu-baseline --side-steps 0 --streams $st --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('streams $st steps 200', 'value %.4e'%d['value'], 'ms/step %.4f'%d['ms_per_step'], d['verified'])
"
python bench.py --no-cpu-baseline --side-steps 0 --streams $st --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('streams $st steps 20 ', 'value %.4e'%d['value'], 'ms/step %.4f'%d['ms_per_step'], d['verified'])
"
done; done
