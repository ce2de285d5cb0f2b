"""Where the CPU leg's time goes on this box: thread counts × phases (diagnostic for bench.py's cpu_baseline)."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from oracle import c_oracle as CO
CO.build()
avail = CO.num_threads()
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for t in sorted({1, max(1, avail // 8), max(1, avail // 4), max(1, avail // 2), avail}):
    CO.set_num_threads(t)
    z = np.zeros(ns * ns)
    CO.bratu_newton_fast(ns, 6.0, 0.0, z, 1, True, 30)
    k = 1 if t == 1 else 4
    _, _, sec = CO.bratu_newton_fast(ns, 6.0, 0.0, z, k, True, 30)
    ph = CO.phase_times() / k
    print(json.dumps(dict(threads=t, steps_per_s=round(k / sec, 3), ms_per_step=round(1e3 * sec / k, 2),
                          phases_ms=dict(zip(["operator", "dots", "reduce_tail_wait", "axpy", "rest"], np.round(1e3 * ph, 2).tolist())),
                          triad=round(CO.stream_triad(1 << 25, 2), 1))), flush=True)
