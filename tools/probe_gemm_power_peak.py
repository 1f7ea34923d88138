"""What dense MFMA rate does the VENDOR's GEMM sustain on this board under the 1400-W cap?  (the practical, power-capped ceiling next to
the 2.5 PFLOP/s nominal fp16 / bf16 peak the bench's roofline.frac is quoted against).  torch.matmul (hipBLASLt), several seconds each."""
import json
import time

import torch

for dt in (torch.float16, torch.bfloat16):
    for n in (4096, 8192, 16384):
        a = torch.randn(n, n, device="cuda", dtype=dt)
        b = torch.randn(n, n, device="cuda", dtype=dt)
        for _ in range(20):
            a @ b
        torch.cuda.synchronize()
        reps = max(20, int(3.0 / (2 * n ** 3 / 1.2e15)))
        t0 = time.perf_counter()
        for _ in range(reps):
            a @ b
        torch.cuda.synchronize()
        dtm = (time.perf_counter() - t0) / reps
        print(json.dumps({"dtype": str(dt), "n": n, "reps": reps, "ms": dtm * 1e3, "tflops": 2 * n ** 3 / dtm / 1e12}), flush=True)
