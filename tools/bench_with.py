#!/usr/bin/env python
"""Run bench.py with plan-structure constants of sipmask_amd.engine changed first -- the A/B route for the switches that
are module constants rather than environment variables:

    python tools/bench_with.py _K32_RING=2 _FUSED_MASKS=False -- --steps 50 --no-cpu-baseline --no-extras

The `other_configs` children of bench.py are separate processes and do not see the change.
"""
import ast
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    argv = sys.argv[1:]
    cut = argv.index("--") if "--" in argv else len(argv)
    import sipmask_amd.engine as E
    for item in argv[:cut]:
        name, value = item.split("=", 1)
        if not hasattr(E, name):
            raise SystemExit("sipmask_amd.engine has no constant %s" % name)
        setattr(E, name, ast.literal_eval(value))
    sys.argv = [os.path.join(ROOT, "bench.py")] + argv[cut + 1:]
    import bench
    bench.main()


if __name__ == "__main__":
    main()
