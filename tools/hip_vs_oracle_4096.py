"""HIP vs the CPU oracle on the 4096^2 golden scenario (tests/golden/big_step2_4096): max and quantiles of |difference| / max|field|."""
import sys
import numpy as np
sys.path.insert(0, "webgl-fluid-simulation_amd"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import scenario as S
from oracle import oracle as O
g, sc = S.load("big_step2_4096")
ref = S.OracleAdapter(O, S.canvas_of(g), sc.get("config"), sc.get("seed", 1234))
want, _ = S.replay(ref, g, sc)
for schedule in ("fused", "passes"):
    ad = S.HipAdapter(S.canvas_of(g), sc.get("config"), sc.get("seed", 1234), schedule=schedule)
    out, _ = S.replay(ad, g, sc)
    ad.close()
    for k in S.FIELDS:
        d = np.abs(out[k].astype(np.float64) - want[k]).ravel() / float(np.abs(want[k]).max())
        qs = np.quantile(d, [0.5, 0.99, 0.999, 0.9999, 0.99999])
        print(schedule, k, "max %.2e" % d.max(), "q50/99/99.9/99.99/99.999:", " ".join("%.1e" % q for q in qs), "frac>6e-5: %.2e" % (d > 6e-5).mean())
