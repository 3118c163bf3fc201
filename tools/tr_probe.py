"""Which value does each lane hold after the resident kernel's transposing reduction? (diagnostic)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from invcompcamtrack_amd import _lib
L = _lib.load()
rng = np.random.default_rng(5)
vals = rng.integers(-500, 500, (64, 64)).astype(np.float32)
out = np.zeros(64, np.float32)
pl, kl = np.zeros(64, np.int32), np.zeros(64, np.int32)
_lib.check(L.ictr_debug_transpose_reduce(_lib.fp(vals), _lib.fp(out), pl.ctypes.data_as(_lib.IP), kl.ctypes.data_as(_lib.IP), int(sys.argv[1]) if len(sys.argv) > 1 else 32))
tot = vals.sum(axis=0)
for l in range(64):
    hit = np.flatnonzero(tot == out[l])
    exp = 2 * pl[l] + kl[l]
    print(f"lane {l:2d}: out {out[l]:9.1f} expected value {exp:2d} (patch {pl[l]:2d} kind {kl[l]}) sum {tot[exp]:9.1f}  actual value(s) {hit.tolist()}"
          + ("" if (len(hit) and hit[0] == exp) else "   <-- MISMATCH"))
