"""TEST INFRASTRUCTURE ONLY -- puts the product's two drop-in modules where the reference looks for its Cython extensions,
exactly as INTEGRATION.md section 1 prescribes:

    sys.modules['CRISPResso2.CRISPResso2Align']        = crispresso2_amd.CRISPResso2Align
    sys.modules['CRISPResso2.CRISPRessoCOREResources'] = crispresso2_amd.CRISPRessoCOREResources

with `CRISPResso2` itself a namespace over the reference's UNMODIFIED source directory (its __init__ imports the plotting
stack, which is not installed here, so the package object is made by hand; seaborn is stubbed; importlib.metadata answers the
version lookup of CRISPRessoShared.py:35-39).  The reference's CRISPRessoCORE / CRISPRessoShared / plots.data_prep then import
the shim through their own, unchanged import statements (CRISPRessoCORE.py:25,32; CRISPRessoShared.py:31-32).

Device: with C2_DROPIN_DEVICE=emulator (default when no GPU is visible) the shim's three C-ABI calls -- c2_global_align,
c2_find_indels_substitutions, c2_calculate_homology -- go to the wave emulator's entry points of the same signatures
(tests/emu: the same HIP kernel source compiled for the host); with a GPU they go to libcrispresso2_amd.so.

Also a pytest plugin (`-p dropin_inject`): the reference's own unit-test files, collected unchanged, then run against the shim."""
import importlib
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("C2_REFERENCE_DIR", "/root/reference")
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


class _EmuLib:
    def __init__(self):
        import emu_driver as E
        lib = E.lib()
        self.c2_global_align = lib.emu_c2_global_align
        self.c2_find_indels_substitutions = lib.emu_c2_find_indels_substitutions
        self.c2_calculate_homology = lib.emu_c2_calculate_homology

    @staticmethod
    def c2_consensus_pairs_batch(handle, *args):                    # (crispresso2_amd.prime's consensus of the registered pairs)
        from pipeline_on_emulator import EmulatedContext
        return EmulatedContext().lib.c2_consensus_pairs_batch(handle, *args)


class _EmuContext:
    """stands in for _native.Context for the per-call API: same attributes the shim modules use (plus the batched classifier that
    crispresso2_amd.prime asks for)"""
    handle = None

    def __init__(self):
        self.lib = _EmuLib()

    def classify_lists_batch(self, *a, **k):
        from pipeline_on_emulator import EmulatedContext
        return EmulatedContext().classify_lists_batch(*a, **k)

    @staticmethod
    def check(rc, what):
        if rc != 0:
            from crispresso2_amd import _native
            raise _native.NativeError("%s failed on the emulator (rc=%d)" % (what, rc))


def use_emulator():
    want = os.environ.get("C2_DROPIN_DEVICE", "")
    if want:
        return want == "emulator"
    try:
        import torch
        return not torch.cuda.is_available()
    except ImportError:
        return True


_done = {}


def emulator_context():
    """the emulator in the place of the GPU context (also what a forked worker's helper process preloads:
    C2_HELPER_PRELOAD=dropin_inject:emulator_context).  "Opening" it marks the process as one that has opened the GPU, so that
    a fork()ed child treats it exactly as it must treat a HIP context: not at all."""
    from crispresso2_amd import _native, batch
    from pipeline_on_emulator import EmulatedAligner
    ctx = _EmuContext()
    _native.default_context = lambda *a, **k: ctx
    _native.note_gpu_opened()
    batch.BatchAligner = EmulatedAligner                             # (crispresso2_amd.prime's one-batch-per-amplicon route)
    os.environ["C2_HELPER_PRELOAD"] = "dropin_inject:emulator_context"
    os.environ["PYTHONPATH"] = HERE + os.pathsep + os.environ.get("PYTHONPATH", "")
    return ctx


def inject():
    """-> (shim align module, shim resources module); idempotent"""
    if _done:
        return _done["A"], _done["R"]
    import importlib.metadata as md
    from crispresso2_amd import CRISPResso2Align as A, CRISPRessoCOREResources as R, _native
    if use_emulator():
        emulator_context()
    pkg = types.ModuleType("CRISPResso2")
    pkg.__path__ = [os.path.join(REF, "CRISPResso2")]
    sys.modules["CRISPResso2"] = pkg
    sys.modules["CRISPResso2.CRISPResso2Align"] = A
    pkg.CRISPResso2Align = A
    sys.modules["CRISPResso2.CRISPRessoCOREResources"] = R
    pkg.CRISPRessoCOREResources = R
    if "seaborn" not in sys.modules:
        sb = types.ModuleType("seaborn")
        sb.set_context = sb.set = sb.set_style = sb.set_theme = lambda *a, **k: None
        sb.matrix = types.SimpleNamespace(_HeatMapper=object)
        sb.utils = types.SimpleNamespace()
        sys.modules["seaborn"] = sb
    orig = md.version
    md.version = lambda name: "2.3.4" if name.lower().startswith("crispresso") else orig(name)
    try:
        import pytest_check  # noqa: F401
    except ImportError:
        # the reference's test_CRISPRessoCORE.py uses the pytest-check plugin (not installed here): a strict stand-in whose
        # checks fail at once instead of being collected
        class _Check:
            def __enter__(self): return self
            def __exit__(self, *exc): return False
            @staticmethod
            def equal(a, b, msg=""): assert a == b, msg
            @staticmethod
            def is_true(x, msg=""): assert bool(x), msg
            @staticmethod
            def is_false(x, msg=""): assert not bool(x), msg
        pc = types.ModuleType("pytest_check")
        pc.check = _Check()
        sys.modules["pytest_check"] = pc
    try:
        import inline_snapshot  # noqa: F401
    except ImportError:
        # likewise inline-snapshot: `assert value == snapshot(recorded)` compares with the recorded value
        isn = types.ModuleType("inline_snapshot")
        isn.snapshot = lambda recorded=None: recorded
        sys.modules["inline_snapshot"] = isn
    _done["A"], _done["R"] = A, R
    return A, R


def reference_core():
    """the reference's unmodified CRISPRessoCORE module, importing the shim"""
    inject()
    return importlib.import_module("CRISPResso2.CRISPRessoCORE")


def run_core_main(argv):
    """CRISPRessoCORE.main() with sys.argv = argv (a SystemExit with code 0 / None is the normal end)"""
    core = reference_core()
    saved = sys.argv
    sys.argv = list(argv)
    try:
        core.main()
    except SystemExit as e:
        if e.code not in (0, None):
            raise
    finally:
        sys.argv = saved


# pytest plugin: inject before the reference's test modules are imported
def pytest_configure(config):
    inject()
