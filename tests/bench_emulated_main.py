"""TEST INFRASTRUCTURE ONLY -- `python tests/bench_emulated_main.py <bench.py arguments>`: bench.py's main() with its device calls on the
wave emulator (bench_on_emulator), as a command.  Started with --gpus N and no RANK / WORLD_SIZE in the environment it goes through
bench.py's own launcher (_spawn_ranks), whose ranks are this script again -- the bare `python bench.py --gpus N` of the driver, on a
machine without GPUs (C2_BENCH_BACKEND=gloo).  A rank prints `RESULT <json or null>`."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p_ in (HERE, ROOT):
    if p_ not in sys.path:
        sys.path.insert(0, p_)

import bench                          # noqa: E402
import bench_on_emulator as BE        # noqa: E402

bench.SPAWN_CMD = [sys.executable, os.path.abspath(__file__)]
out = BE.run_bench(sys.argv[1:])
print("RESULT " + json.dumps(out))
