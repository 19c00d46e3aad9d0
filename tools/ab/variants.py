#!/usr/bin/env python3
"""Same-box A/B/C... of builds and debug knobs of the library: runs bench.py (2 M reads by default, no CPU leg, no checks) for every
variant, alternating, `--rounds` times, and prints per variant the dominant kernel's average launch, the chain and the step.
  python tools/ab/variants.py [--reads N] [--steps K] [--rounds R] [--config C] label[=path/to/lib.so][,ENV=VAL...] ...
A variant without a path uses the tree's build.  Run on the GPU box (gpurun)."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=2_000_000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    res = {}
    for rnd in range(a.rounds):
        for v in a.variants:
            parts = v.split(",")
            label, _, lib = parts[0].partition("=")
            env = dict(os.environ)
            if lib:
                env["C2_AMD_LIB"] = os.path.join(ROOT, lib) if not os.path.isabs(lib) else lib
            for kv in parts[1:]:
                k, _, val = kv.partition("=")
                env[k] = val or "1"
            p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", str(a.config), "--reads", str(a.reads), "--steps", str(a.steps),
                                "--warmup", "1", "--no-cpu-baseline", "--check", env.get("C2_AB_CHECK", "0"), "--no-full-plane-check", "--no-dedup-leg", "--workers", "8"],
                               capture_output=True, text=True, env=env, cwd=ROOT)
            line = [x for x in p.stdout.splitlines() if x.startswith("{")]
            if not line:
                print(label, "FAILED", p.stderr[-400:])
                continue
            d = json.loads(line[-1])
            r = d["roofline"]
            row = (r["avg_launch_ms"], r["chain_avg_ms"], d["ms_per_step"], d["step_breakdown_ms"]["count_vectors_and_all_reduce"])
            res.setdefault(label, []).append(row)
            print("%-28s first kernel %8.3f ms  chain %8.3f ms  step %8.3f ms  count %7.3f ms  left %s  %s" %
                  ((label,) + row + (d["config"]["tasks_left_after_each_banded_launch"], d.get("checks", ""))), flush=True)
    print(json.dumps({k: [sum(x[i] for x in v) / len(v) for i in range(4)] for k, v in res.items()}))


if __name__ == "__main__":
    main()
