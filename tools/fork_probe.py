"""What a fork()ed child of a process that has opened the GPU can do with HIP ("prove it on the GPU box").
The parent opens a context and aligns once; each child tries ONE thing with a 25 s alarm and reports through a pipe:
  a. use the parent's context (c2_global_align on the inherited handle),
  b. create a context of its own (c2_create),
  c. nothing of HIP: spawn `python -m crispresso2_amd._helper` and ask IT (the route the shim takes).
Prints one JSON line.  crispresso2_amd's own guard is bypassed here (the library is called through ctypes directly)."""
import ctypes
import json
import os
import signal
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crispresso2_amd import CRISPResso2Align as A, _native, synth


def child(what, w, ctx, amp, g, m):
    signal.alarm(25)
    t0 = time.time()
    try:
        if what == "inherited_context":
            bj, bi = amp[20:120].encode(), amp.encode()
            oj, oi = ctypes.create_string_buffer(600), ctypes.create_string_buffer(600)
            n, mt, st = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
            rc = ctx.lib.c2_global_align(ctx.handle, bj, len(bj), bi, len(bi), m.ctypes.data_as(ctypes.c_void_p), int(m.shape[0]),
                                         g.ctypes.data_as(ctypes.c_void_p), int(g.shape[0]), -20, -2, oj, oi, ctypes.byref(n), ctypes.byref(mt), ctypes.byref(st))
            msg = ctx.lib.c2_last_error(ctx.handle)
            out = {"rc": rc, "aln_len": n.value, "error": msg.decode() if msg else None}
        elif what == "own_context":
            h = ctypes.c_void_p()
            rc = ctx.lib.c2_create(0, ctypes.byref(h))
            msg = ctx.lib.c2_last_error(None)
            out = {"rc": rc, "error": msg.decode() if msg else None}
        else:
            got = A.global_align(amp[20:120], amp, matrix=m, gap_incentive=g, gap_open=-20, gap_extend=-2)
            out = {"rc": 0, "aln_len": len(got[0]), "helper_calls": _native._helper[1].calls}
    except BaseException as e:
        out = {"exception": "%s: %s" % (type(e).__name__, e)}
    out["seconds"] = round(time.time() - t0, 2)
    os.write(w, json.dumps(out).encode())
    os._exit(0)


def main():
    amp, g, inc = synth.amplicon_setup(250)
    g = np.ascontiguousarray(g, dtype=np.int64)
    m = A.read_matrix(os.path.join(os.path.dirname(A.__file__), "EDNAFULL"))
    first = A.global_align(amp[20:120], amp, matrix=m, gap_incentive=g, gap_open=-20, gap_extend=-2)
    ctx = _native.default_context()
    res = {"parent_first": len(first[0])}
    for what in ("shim_helper", "own_context", "inherited_context"):
        r, w = os.pipe()
        pid = os.fork()
        if pid == 0:
            os.close(r)
            child(what, w, ctx, amp, g, m)
        os.close(w)
        t0 = time.time()
        done = False
        while time.time() - t0 < 40:
            p, status = os.waitpid(pid, os.WNOHANG)
            if p:
                done = True
                break
            time.sleep(0.1)
        if not done:
            os.kill(pid, signal.SIGKILL)
            os.waitpid(pid, 0)
            res[what] = {"hung": True}
        else:
            data = os.read(r, 65536)
            res[what] = json.loads(data) if data else {"died": "signal %d" % (status & 0x7f) if status & 0x7f else "exit %d" % (status >> 8)}
        os.close(r)
    again = A.global_align(amp[20:120], amp, matrix=m, gap_incentive=g, gap_open=-20, gap_extend=-2)
    res["parent_after_forks_ok"] = again == first
    print(json.dumps(res))


if __name__ == "__main__":
    main()
