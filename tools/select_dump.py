"""Writes the inputs of tools/select_bench.cpp (C2 scene/model/keys) to a binary file."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hop_loader
hop = hop_loader.load()
synth = hop.synth
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/sel_dump.bin"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
sc = synth.make_scene(n, seed=7)
mx, mn = synth.ellipsoid_model(5000)
keys = synth.ppf_key_table()
with open(out, "wb") as f:
    np.array([len(sc.xyz), len(mx), len(keys)], np.int32).tofile(f)
    np.ascontiguousarray(sc.xyz.T).tofile(f); np.ascontiguousarray(sc.nrm.T).tofile(f); sc.conf.tofile(f)
    np.ascontiguousarray(mx.T).tofile(f); np.ascontiguousarray(mn.T).tofile(f); keys.astype(np.int32).tofile(f)
print("wrote", out)
