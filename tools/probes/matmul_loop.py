#!/usr/bin/env python3
"""torch.matmul (hipBLASLt) 8192^3 bf16 on N(0, 1) data back to back for ~10 s: the load for tools/power_sample.sh."""
import time
import torch
a = torch.randn(8192, 8192, device="cuda").to(torch.bfloat16)
b = torch.randn(8192, 8192, device="cuda").to(torch.bfloat16)
t0 = time.time()
while time.time() - t0 < 10:
    for _ in range(200):
        torch.matmul(a, b)
    torch.cuda.synchronize()
