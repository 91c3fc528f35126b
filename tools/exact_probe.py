import sys, time, struct
sys.path.insert(0, '.')
import __graft_entry__ as ge
import numpy as np
pkg = ge.load_package()
n = 1342177280
g = pkg.PaprHip(0)
g.set_exact(True)
g.generate(pkg.SynthSpec.spike(n), 0, n)
st = g.stats()
mean, papr, table = pkg.levels(st, False)
for it in range(6):
    t0 = time.perf_counter(); st = g.stats(); t1 = time.perf_counter()
    counts, prog = g.ccdf_exact(table, 0.0, n); t2 = time.perf_counter()
    s = pkg.exact_chain([prog]); t3 = time.perf_counter()
    hdr = struct.unpack_from("<IIQQQIIII", prog, 0)
    print(f"stats {1e3*(t1-t0):.3f} ms  ccdf_exact {1e3*(t2-t1):.3f} ms  chain {1e3*(t3-t2):.3f} ms  program {len(prog)} B groups={hdr[4]} tail={hdr[5]} mixed={hdr[6]} raw={hdr[7]}")
