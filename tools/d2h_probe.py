import time, torch
x = torch.randn(15, 3_000_000, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for name in ("pageable .cpu()", "pinned fresh", "pinned cached"):
    for rep in range(3):
        t0 = time.perf_counter()
        if name.startswith("pageable"):
            h = x.cpu()
        else:
            h = torch.empty(x.shape, dtype=x.dtype, pin_memory=True)
            h.copy_(x, non_blocking=True); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name} rep {rep}: {dt*1e3:.1f} ms  {x.numel()*8/dt/1e9:.1f} GB/s")
        if name == "pinned fresh": del h; torch.cuda.empty_cache()
import numpy as np, pandas as pd
h = torch.empty(x.shape, dtype=x.dtype, pin_memory=True); h.copy_(x); 
t0=time.perf_counter(); df = pd.DataFrame(h.numpy().T, copy=False); print("frame", (time.perf_counter()-t0)*1e3, "ms", df.shape)
