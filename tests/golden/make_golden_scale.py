#!/usr/bin/env python3
"""Record the REAL reference's stdout for the multi-GPU bench workloads (build container only).

bench.py --gpus N runs N x 10 GiB of the synthetic spike stream (weak scaling; N = 8 is BASELINE.json
configs[3], 80 GiB).  For every N given this script writes that stream to a scratch file with oracle/mkcfile,
runs oracle/_ref/papr (compiled from /root/reference/papr.c where it lies) on it in both modes and stores

    tests/golden/big_spike<N*10>g.default.txt / .graph.txt     the reference's stdout
    manifest.json["big_spike<N*10>g"]                           sizes, exit codes, sha256 of both stdouts,
                                                                the oracle's sequential sum (hex)

so that bench.py can compare what the GPUs print with what the reference printed for the very same stream
("parity_in_run") at every N the driver benches, and the GPU tests can do the same through the ABI.
Only data (expected outputs) is stored; the scratch file is deleted.

    python tests/golden/make_golden_scale.py --scratch /root/scratch 1 2 4 8
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
ORACLE = os.path.join(ROOT, "oracle")
MKCFILE = os.path.join(ORACLE, "mkcfile")
REF = os.path.join(ORACLE, "_ref", "papr")
PER_GPU = 1342177280  # samples in 10 GiB


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("ngpus", nargs="+", type=int)
    ap.add_argument("--scratch", default="/tmp")
    ap.add_argument("--skip-graph", action="store_true")
    args = ap.parse_args()
    subprocess.check_call(["make", "-C", ORACLE, "all"])
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/papr missing: this script needs /root/reference")
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    orc = ge.load_oracle()
    os.makedirs(args.scratch, exist_ok=True)
    manifest_path = os.path.join(HERE, "manifest.json")
    for n_gpu in args.ngpus:
        n = PER_GPU * n_gpu
        name = f"big_spike{10 * n_gpu}g"
        path = os.path.join(args.scratch, name + ".cfile")
        t0 = time.time()
        subprocess.check_call([MKCFILE, path, str(n), "--spike"])
        print(f"{name}: wrote {n * 8 >> 30} GiB in {time.time() - t0:.0f} s", flush=True)
        entry = {"bytes": n * 8, "nsamples": n, "synthetic": "spike", "ranks": n_gpu}
        try:
            for graph, tag in ((False, "default"), (True, "graph")):
                if graph and args.skip_graph:
                    continue
                t0 = time.time()
                p = subprocess.run([REF] + (["-g"] if graph else []) + [path], capture_output=True)
                with open(os.path.join(HERE, f"{name}.{tag}.txt"), "wb") as f:
                    f.write(p.stdout)
                entry[tag] = {"rc": p.returncode, "stderr": p.stderr.decode(), "lines": p.stdout.count(b"\n"),
                              "sha256": hashlib.sha256(p.stdout).hexdigest(), "reference_wall_s": round(time.time() - t0, 1)}
                print(f"{name}: reference {tag} {time.time() - t0:.0f} s", flush=True)
            t0 = time.time()
            r = orc.run_file(path, False)
            entry["oracle_sequential_sum_hex"] = float(r["sum"]).hex()
            print(f"{name}: oracle {time.time() - t0:.0f} s, sum {entry['oracle_sequential_sum_hex']}", flush=True)
        finally:
            os.unlink(path)
        manifest = json.load(open(manifest_path))
        old = manifest.get(name, {})
        old.update(entry)
        manifest[name] = old
        json.dump(manifest, open(manifest_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
