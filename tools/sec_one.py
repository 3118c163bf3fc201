"""Run selected secondary records: python tools/sec_one.py c3 c5 ..."""
import sys, json
sys.path.insert(0, ".")
from tools import secondary
for name in sys.argv[1:]:
    r = getattr(secondary, "rec_" + name)(1.0)
    print(json.dumps({k: v for k, v in r.items() if k in ("name", "value", "unit", "ms_per_step", "kernel_us", "frac")}))
