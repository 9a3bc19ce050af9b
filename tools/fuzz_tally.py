"""How often the random benches of tests/test_gpu_fuzz.py run a generation dense with its absorbed rays kept (hint mode 4)
or dense:  python tools/fuzz_tally.py [first] [count]"""
import sys, collections
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import test_gpu_fuzz as f
from pyrayt_amd import engine
tally = collections.Counter()
close = engine.DeviceScene.close
def closing(self):
    told = self.telemetry()
    tally["scenes"] += 1
    tally["scenes_with_keep"] += told["sparse_keep_launches"] > 0
    tally["keep_launches"] += told["sparse_keep_launches"]
    tally["dense_launches"] += told["dense_launches"]
    close(self)
engine.DeviceScene.close = closing
first = int(sys.argv[1]) if len(sys.argv) > 1 else 900000
for seed in range(first, first + (int(sys.argv[2]) if len(sys.argv) > 2 else 400)):
    f.test_random_bench(seed)
print(dict(tally))
