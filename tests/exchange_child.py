"""Body of tests/test_gpu_fullsize.py::test_device_resident_topk_exchange_and_frame_gather_single_rank, run as a process of its own."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hop_loader  # noqa: E402

hop = hop_loader.load()
from hop_amd import api  # noqa: E402

synth = hop.synth
EMU = bool(os.environ.get("HOP_TEST_EMU"))
if EMU:
    # TEST INFRASTRUCTURE (tests/emu): the CPU model of the kernels and its file-based stand-in for RCCL (the parent test put the stand-in on
    # LD_LIBRARY_PATH); "device memory" is host memory there
    api.LIB_PATH = os.environ.get("HOP_TEST_EMU_LIB") or os.path.join(ROOT, "tests", "emu", "_build", "libhop_emu.so")
    api._lib = None
api.lib()
ctx = api.Context(0)
if EMU:
    _host_rows = np.zeros((128, api.TOPK_ROW_FLOATS), np.float32)
    dbuf = C.c_void_p(_host_rows.ctypes.data)

    class hip:  # noqa: N801
        @staticmethod
        def hipFree(_):
            return 0

    def dev_rows():
        ctx.synchronize()
        return _host_rows.copy()
else:
    hip = C.CDLL("libamdhip64.so")     # a 9 KB device buffer without PyTorch (whose bundled HIP / RCCL must not mix with the system's here)
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    dbuf = C.c_void_p()
    assert hip.hipMalloc(C.byref(dbuf), 128 * api.TOPK_ROW_FLOATS * 4) == 0

    def dev_rows():
        out = np.zeros((128, api.TOPK_ROW_FLOATS), np.float32)
        assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), dbuf, out.nbytes, 2) == 0   # hipMemcpyDeviceToHost
        return out


sc = synth.make_scene(2000, seed=7)
poses = synth.replay_poses(sc.gt_pose, 500, seed=1)
scores = np.random.default_rng(0).random(500).astype(np.float32)
scores[10] = scores[20] = scores[30]
ctx.hypos_upload(poses, scores)
rows_h, n = ctx.topk_pack(128, id_offset=1000)
p, s, i = ctx.hypos_download()
order = np.lexsort((i, -s))[:128]
assert n == 128 and np.array_equal(rows_h[:, 0], s[order]) and np.array_equal(rows_h[:, 1].copy().view(np.int32), i[order] + 1000)
assert np.array_equal(rows_h[:, 2:], p[order].reshape(-1, 16))
try:
    comm = api.Comm(0, api.Comm.unique_id(), 0, 1)
except api.HopError as e:
    print(f"SKIP RCCL not loadable here: {e}")
    sys.exit(0)
try:
    assert ctx.topk_pack_device(128, 1000, dbuf.value) == 128
    assert np.array_equal(dev_rows(), rows_h)
    m, nm = comm.topk_allgather_device(dbuf.value, 128)
    m2, _ = comm.topk_allgather(rows_h, 128)
    assert nm == 128 and np.array_equal(m, m2) and np.array_equal(m, rows_h)
    ctx.hypos_upload(poses[:5], scores[:5])         # fewer hypotheses than k: padding rows carry id -1 and sort last
    ctx.topk_pack_device(128, 0, dbuf.value)
    m, nm = comm.topk_allgather_device(dbuf.value, 128)
    assert nm == 5 and (m[5:, 1].copy().view(np.int32) == -1).all() and (np.diff(m[:5, 0]) <= 0).all()
    fr = np.zeros((3, api.FRAME_ROW_FLOATS), np.float32)
    fr[:, 0] = [4, 9, 2]
    fr[:, 1:] = np.random.default_rng(1).normal(size=(3, 16))
    g = comm.frames_allgather(fr, 5, 1)
    assert np.array_equal(g[:3], fr) and (g[3:, 0] == -1).all()
    nr, us, cnt = comm.info()
    assert nr == 1 and cnt == 3
finally:
    comm.close()
    hip.hipFree(dbuf)
ctx.close()
print("EXCHANGE OK")
