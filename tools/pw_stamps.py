#!/usr/bin/env python
"""(needs the stamps build: make -C nonlinearsolve.jl_amd/csrc stamps; NK_LIB_PATH=nonlinearsolve.jl_amd/lib/libmi355x_nk_stamps.so)
Development: where the time of ONE POWER of the resident matrix-powers kernel goes — phase stamps of every band's wavefront 0
(100 MHz wall clock, 10 ns resolution), Bratu 1024² (256 bands of 4 slices, W = 5), 15 powers per launch, the steady-state powers
2 … 13 of the last of `reps` launches. NK_PW_GRAN=0 / 1 selects the hand-off form (a process each).

    python tools/pw_stamps.py [grid=1024] [s=15] [reps=20]
"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import nonlinearsolve_jl_amd as nls
from nonlinearsolve_jl_amd import _lib as L

NST, STP = 12, 16
f = L.lib().nk_pw_debug_stamps
f.argtypes = [C.c_int, C.POINTER(C.c_ulonglong), C.c_int]
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
s = int(sys.argv[2]) if len(sys.argv) > 2 else 15
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
P = nls.Bratu2D(ns, 6.0)
n = ns * ns
u = torch.zeros(n, dtype=torch.float64, device="cuda")
J = P.jac_csr()
P.jac_values(u, J)
x = torch.randn(n, dtype=torch.float64, device="cuda")
lam = 8.0 * (ns + 1) ** 2
theta = (0.5 + 0.4 * np.cos(np.arange(s))) * lam
f(1, None, 0)
for _ in range(reps):
    Y, resident = J.powers(x, s, theta=theta, scale=2.0 / lam)
torch.cuda.synchronize()
nb = (n + 4095) // 4096
out = (C.c_ulonglong * (nb * STP * NST))()
f(1, out, nb)
st = np.array(list(out), dtype=np.float64).reshape(nb, STP, NST) / 100.0   # µs
gran = os.environ.get("NK_PW_GRAN", "1") != "0"
names = {0: "power starts", 1: "boundary slices computed, stores issued", 2: "write-through stores drained (vmcnt 0)",
         3: "barrier behind the drain", 4: "interior slices computed, stores issued", 5: "upper neighbour's flag seen",
         6: "both flags seen (barrier)", 7: "halo rows / granules in (this wavefront)", 9: "power ends (barrier)"}
ps = [p for p in range(2, min(s - 1, STP) - 1)]
rows = []
print(f"# resident matrix-powers kernel, one power, by phase — Bratu {ns}², {nb} bands, s = {s}, hand-off: "
      f"{'16-byte {value, tag} granules' if gran else 'sc1 payload -> drain -> flag'} (resident: {bool(resident)})")
print(f"# µs from the power's start, wavefront 0 of every band, powers {ps[0]} … {ps[-1]} of the last launch: median [p10, p90] over bands × powers")
order = [1, 2, 3, 4, 5, 6, 7, 9]
prev = None
for i in order:
    d = np.array([st[b, p, i] - st[b, p, 0] for b in range(1, nb - 1) for p in ps if st[b, p, i] > 0 and st[b, p, 0] > 0])
    if d.size == 0:
        continue
    med, p10, p90 = np.median(d), np.percentile(d, 10), np.percentile(d, 90)
    print(f"| {names[i]:48s} | {med:6.2f} | [{p10:5.2f}, {p90:5.2f}] | +{med - (prev or 0.0):5.2f} |")
    prev = med
tot = np.array([st[b, p + 1, 0] - st[b, p, 0] for b in range(1, nb - 1) for p in ps if st[b, p + 1, 0] > 0])
print(f"| {'one power (start to next start)':48s} | {np.median(tot):6.2f} | [{np.percentile(tot, 10):5.2f}, {np.percentile(tot, 90):5.2f}] |")
whole = np.array([st[b, s - 1, 4] - st[b, 0, 0] for b in range(nb) if st[b, s - 1, 4] > 0])
print(f"| {'powers 0 … s−1 (first start to last compute)':48s} | {np.median(whole):6.2f} |")
# bands whose neighbour sits on another XCD (the band → XCD map is 32 consecutive bands per XCD at 256 bands)
q = nb // 8
if q > 1:
    edge = [b for b in range(1, nb - 1) if (b % q) in (0, q - 1)]
    inner = [b for b in range(1, nb - 1) if (b % q) not in (0, q - 1)]
    for nm, bs in (("bands with both neighbours on their XCD", inner), ("bands next to an XCD boundary", edge)):
        d = np.array([st[b, p + 1, 0] - st[b, p, 0] for b in bs for p in ps if st[b, p + 1, 0] > 0])
        print(f"| {nm:48s} | {np.median(d):6.2f} | [{np.percentile(d, 10):5.2f}, {np.percentile(d, 90):5.2f}] |")
print(json.dumps(dict(grid=ns, s=s, granules=gran, us_per_power=round(float(np.median(tot)), 3), checksum=float(Y[-1].abs().sum().item()))))
