"""`python -m crispresso2_amd._helper <read fd> <write fd>`: the per-call entry points of the two drop-in modules, served to a
process that cannot open the GPU itself -- a fork()ed worker of the reference's `-p N` route (CRISPRessoCORE.py:1870-1898,
:1198-1242), whose parent had already initialised HIP when it forked.  This process is SPAWNED (fork + exec), so its HIP runtime
is its own.  Requests: length-prefixed pickles (name, args); answers ("ok", value) or ("err", exception).  A forked worker whose
parent primed its reads before the fork (crispresso2_amd.prime) never starts one.

C2_HELPER_PRELOAD=module:function is called first (a host that installs its own context -- the tests' wave emulator)."""
import os
import pickle
import struct
import sys


def main(rfd, wfd):
    r = os.fdopen(rfd, "rb", buffering=0)
    w = os.fdopen(wfd, "wb", buffering=0)
    pre = os.environ.get("C2_HELPER_PRELOAD")
    if pre:
        import importlib
        mod, fn = pre.split(":")
        getattr(importlib.import_module(mod), fn)()
    from crispresso2_amd import CRISPResso2Align as A, CRISPRessoCOREResources as R, _native
    table = {"global_align": A.global_align, "find_indels_substitutions": R.find_indels_substitutions,
             "find_indels_substitutions_legacy": R.find_indels_substitutions_legacy, "calculate_homology": R.calculate_homology,
             "ping": lambda: os.getpid()}
    while True:
        head = _native._read_exactly(r, 8)
        if head is None:
            return
        name, args = pickle.loads(_native._read_exactly(r, struct.unpack("<Q", head)[0]))
        try:
            out = ("ok", table[name](*args))
        except BaseException as e:                                     # the caller re-raises it in the worker
            try:
                pickle.dumps(e)
                out = ("err", e)
            except Exception:
                out = ("err", Exception("%s: %s" % (type(e).__name__, e)))
        blob = pickle.dumps(out, protocol=pickle.HIGHEST_PROTOCOL)
        w.write(struct.pack("<Q", len(blob)) + blob)


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]))
