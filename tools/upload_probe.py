import os, sys, time, tempfile, threading
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
from crispresso2_amd import synth
reads = synth.make_reads(250, 10_000_000, workers=32)
d = tempfile.mkdtemp(prefix="c2up_", dir="/dev/shm"); p = os.path.join(d, "r.fastq")
synth.write_fastq(reads, p); del reads
n = os.path.getsize(p)
dev = torch.device("cuda", 0)
d_text = torch.empty(n, dtype=torch.uint8, device=dev)
CH = 256 << 20
pins = [torch.empty(CH, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
copy_stream = torch.cuda.Stream()
for threads in (8, 16, 32):
    for rep in range(2):
        mm = np.memmap(p, dtype=np.uint8, mode="r")
        pool = ThreadPoolExecutor(threads)
        evs = [None, None]
        torch.cuda.synchronize(); t0 = time.perf_counter(); t_copy = 0.0
        for c, a in enumerate(range(0, n, CH)):
            k = c & 1; m = min(CH, n - a)
            if evs[k] is not None: evs[k].synchronize()
            dst = pins[k].numpy()
            tc = time.perf_counter()
            step = (m + threads - 1) // threads
            list(pool.map(lambda q: np.copyto(dst[q:min(m, q + step)], mm[a + q:a + min(m, q + step)]), range(0, m, step)))
            t_copy += time.perf_counter() - tc
            with torch.cuda.stream(copy_stream):
                d_text[a:a + m].copy_(pins[k][:m], non_blocking=True)
                evs[k] = torch.cuda.Event(); evs[k].record(copy_stream)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        pool.shutdown(); del mm
        print("threads %d: upload %.3f s (%.1f GB/s), host copies %.3f s" % (threads, dt, n / dt / 1e9, t_copy), flush=True)
# GPU-side: count newlines with torch (cost estimate of a scan)
torch.cuda.synchronize(); t0 = time.perf_counter()
nl = int((d_text == 10).sum().item())
torch.cuda.synchronize(); print("torch newline count %.4f s, %d" % (time.perf_counter() - t0, nl))
os.remove(p); os.rmdir(d)
