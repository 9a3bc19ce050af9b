import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from pyrayt_amd import engine
lib = engine.library()
dev = torch.device("cuda", 0)
sums = torch.rand((4, 9), dtype=torch.float64, device=dev)
sums[:, 0] = 10; sums[:, 8] = 10
piv = torch.zeros((4, 3), dtype=torch.float64, device=dev)
out_dev = torch.empty((4, 8), dtype=torch.float64, device=dev)
out_pin = torch.zeros((4, 8), dtype=torch.float64).pin_memory()
st = engine._stream_ptr(torch, dev)
engine._check(lib.prt_frame_finish(0, sums.data_ptr(), piv.data_ptr(), 4, out_dev.data_ptr(), st))
engine._check(lib.prt_frame_finish(0, sums.data_ptr(), piv.data_ptr(), 4, out_pin.data_ptr(), st))
torch.cuda.synchronize()
print("kernel wrote pinned host memory through the host pointer:", torch.equal(out_dev.cpu(), out_pin))
for name, fn in (("cpu()", lambda: out_dev.cpu()), ("sync only", lambda: torch.cuda.current_stream(dev).synchronize())):
    t0 = time.perf_counter()
    for _ in range(2000):
        engine._check(lib.prt_frame_finish(0, sums.data_ptr(), piv.data_ptr(), 4, (out_dev if name == "cpu()" else out_pin).data_ptr(), st))
        fn()
    print(name, (time.perf_counter() - t0) / 2000 * 1e6, "us per launch + read-back")
