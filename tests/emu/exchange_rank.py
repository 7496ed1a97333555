"""tests/emu/exchange_rank.py -- TEST INFRASTRUCTURE: one rank of the per-frame top-k exchange (BASELINE configs[4]: "hypothesis-parallel with RCCL
top-k all-reduce") on the CPU model: its share of a hypothesis set with scores goes onto the "device" (hop_hypos_upload), the table is packed
there (hop_topk_pack_device: k_score_keys + radix sort + k_topk_pack), exchanged and merged (hop_topk_allgather_device: ncclAllGather answered by
tests/emu/fake_rccl.cpp, then the LDS bitonic k_topk_merge), and the merged table is printed as hex -- no torch in this process (torch would bring
its own librccl).     python exchange_rank.py <rank> <world> <id file> <k> <cases.npz> <model lib>"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hop_loader  # noqa: E402

hop_loader.load()
from hop_amd import api  # noqa: E402

rank, world, id_file, k, cases, lib = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5], sys.argv[6]
api.LIB_PATH = lib
api._lib = None
if rank == 0:
    uid = api.Comm.unique_id()
    open(id_file + ".tmp", "wb").write(uid)
    os.rename(id_file + ".tmp", id_file)
else:
    t0 = time.time()
    while not os.path.exists(id_file):
        assert time.time() - t0 < 60, "rank 0 did not publish the id"
        time.sleep(0.01)
    uid = open(id_file, "rb").read()
comm = api.Comm(0, uid, rank, world)
assert comm.info()[0] == world
ctx = api.Context(0)
g = np.load(cases)
for c in range(int(g["n_cases"])):
    poses, scores = g[f"poses{c}"], g[f"scores{c}"]
    share = np.array_split(np.arange(len(scores)), world)[rank]          # contiguous shards, ids offset by the shard's start (the bench's strong mode)
    rows = np.zeros((k, api.TOPK_ROW_FLOATS), np.float32)                # "device" memory of the model = this process's memory
    if len(share):
        ctx.hypos_upload(poses[share], scores[share])
        ctx.topk_pack_device(k, int(share[0]), rows.ctypes.data)
    else:
        rows[:, 0] = -3.402823466e+38
        rows[:, 1] = np.array([-1], np.int32).view(np.float32)[0]
    merged, n = comm.topk_allgather_device(rows.ctypes.data, k)
    print("case", c, n, merged.view(np.uint32).tobytes().hex(), rows.view(np.uint32).tobytes().hex(), flush=True)
comm.close()
ctx.close()
