#!/usr/bin/env python3
"""Known byte counts for calibrating FETCH_SIZE / WRITE_SIZE (MI355X_MICROARCH.md: check the counters' units on a kernel whose traffic
is known): a fill of N bytes (N written, nothing read) and a copy of N bytes (N read, N written), N = 4 GiB, run under
`rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`.  tools/pmc_calibrate_summary.py turns the two counter files into factors."""
import torch
N = 4 << 30
a = torch.empty(N, dtype=torch.uint8, device="cuda")
b = torch.empty(N, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
a.fill_(1)            # N bytes written
b.copy_(a)            # N read + N written
torch.cuda.synchronize()
print("bytes", N)
