"""Test entry (tests/test_bench_multi.py): bench.main() on CPU ranks over gloo with the CPU oracle injected as the stripe engine —
what `python bench.py --gpus N` does on GPUs, minus the GPUs.  Started WITHOUT a launcher it goes through bench.py's own self-launch
(`launch_ranks`: torch.distributed.run around this very file), which is the path under test; `python bench.py` itself has no such hook."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "webgl-fluid-simulation_amd"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

if __name__ == "__main__":
    import bench
    from oracle_engine import OracleStripeEngine
    bench.main(sys.argv[1:], engine_factory=OracleStripeEngine, backend="gloo")
