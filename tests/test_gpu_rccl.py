"""First contact with RCCL at world > 1.  The driver's GPU box has ONE MI355X, so two ranks can only meet on the same
device: two spawned processes, both on cuda:0, drive `c2_comm_unique_id` / `c2_comm_init(rank, 2)` / `c2_reduce_counts` through the C ABI -- the
library's own dlopen'd RCCL, no torch.distributed in the data path (the 128-byte id travels through a file).  RCCL may refuse two ranks of one
communicator on one device ("Duplicate GPU detected"); which of the two happens is recorded in gpurun_out/ (copied to profiles/r05/):

  * it works  -> every rank's tensor holds the sum over both ranks (asserted), i.e. the N > 1 branch of c2_reduce_counts has executed;
  * refused   -> the test asserts that the refusal is RCCL's own, clean error out of c2_comm_init on BOTH ranks (no hang, no crash): the world > 1 path up to
                 ncclCommInitRank has executed, and the log says which line of bench.py --gpus 2 would have been next.
Either way neither rank may hang: each runs under its own timeout."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

RANK = r'''
import ctypes, json, os, sys, time
sys.path.insert(0, %(root)r)
rank, world, box = int(sys.argv[1]), 2, sys.argv[2]
import torch
from crispresso2_amd import _native
out = {"rank": rank}
try:
    ctx = _native.Context(0)
    buf = (ctypes.c_uint8 * 128)()
    idf = os.path.join(box, "id.bin")
    if rank == 0:
        ctx.check(ctx.lib.c2_comm_unique_id(buf), "c2_comm_unique_id")
        with open(idf + ".tmp", "wb") as fh:
            fh.write(bytes(buf))
        os.replace(idf + ".tmp", idf)
    else:
        t0 = time.time()
        while not os.path.exists(idf):
            if time.time() - t0 > 60:
                raise RuntimeError("rank 0 never published the communicator id")
            time.sleep(0.05)
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(open(idf, "rb").read())
    rc = ctx.lib.c2_comm_init(ctx.handle, rank, world, buf)
    if rc != 0:
        msg = ctx.lib.c2_last_error(ctx.handle)
        out.update(stage="c2_comm_init", rc=rc, error=msg.decode() if msg else None)
    else:
        t = (torch.arange(5000, dtype=torch.int64, device="cuda") * 3 - 7) * (rank + 1)
        ctx.reduce_counts(t.data_ptr(), t.numel(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        want = (torch.arange(5000, dtype=torch.int64, device="cuda") * 3 - 7) * 3      # rank 0's + rank 1's = (1 + 2) x
        out.update(stage="c2_reduce_counts", rc=0, sum_ok=bool(torch.equal(t, want)))
        ctx.check(ctx.lib.c2_comm_destroy(ctx.handle), "c2_comm_destroy")
except BaseException as e:
    out.update(stage=out.get("stage", "exception"), exception="%%s: %%s" %% (type(e).__name__, e))
print("RCCL_RANK " + json.dumps(out))
'''


@pytest.mark.gpu
def test_two_process_rccl_reduce_through_the_c_abi(tmp_path):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG=os.environ.get("NCCL_DEBUG", "WARN"))
    code = RANK % dict(root=ROOT)
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=150)
        except subprocess.TimeoutExpired:
            p.kill()
            so, se = p.communicate()
            so += "\nRCCL_RANK " + json.dumps({"stage": "timeout"})
        outs.append((p.returncode, so, se))
    ranks = []
    for rc, so, se in outs:
        lines = [x for x in so.splitlines() if x.startswith("RCCL_RANK ")]
        assert lines, (rc, so[-1500:], se[-3000:])
        ranks.append(json.loads(lines[-1][10:]))
    log = {"ranks": ranks, "stderr_tail": [se[-1500:] for _, _, se in outs]}
    out_dir = os.path.join(ROOT, "gpurun_out", "r05_rccl")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "two_process_rccl_one_gpu.json"), "w") as fh:
        json.dump(log, fh, indent=1)
    assert all(r["stage"] != "timeout" for r in ranks), log                # nobody may hang
    if all(r.get("stage") == "c2_reduce_counts" for r in ranks):
        assert all(r["sum_ok"] for r in ranks), log                        # RCCL put two ranks on the one device: the sum is the sum
    else:
        # RCCL's refusal of two ranks on one device: a clean error from c2_comm_init (ncclCommInitRank) on every rank that got there
        assert all(r.get("stage") == "c2_comm_init" and r.get("rc") not in (0, None) and r.get("error") for r in ranks), log
