"""The N>1 path on CPU: two processes, gloo, all-gather of the packed top-k tables and the identical merge
(bench.py's exchange()).  Tables are built from synthetic rows; no GPU, no kernel."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, k, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import hop_loader
    hop_loader.load()
    from hop_amd import api
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(100 + rank)
    rows = np.zeros((k, api.TOPK_ROW_FLOATS), np.float32)
    sc = np.sort(rng.integers(0, 50, k))[::-1].astype(np.float32)
    rows[:, 0] = sc
    rows[:, 1] = (np.arange(k, dtype=np.int32) + rank * (1 << 24)).view(np.float32)
    rows[:, 2:] = rng.normal(size=(k, 16))
    if rank == 1:  # a rank with fewer hypotheses than k pads with id = -1
        rows[k - 5:, 0] = -np.finfo(np.float32).max
        rows[k - 5:, 1] = np.array([-1], np.int32).view(np.float32)[0]
    t = torch.from_numpy(rows)
    out = torch.empty((world * t.shape[0], t.shape[1]), dtype=t.dtype)  # concatenated layout: valid for gloo and nccl
    dist.all_gather_into_tensor(out, t)
    merged, n = api.topk_merge(out.numpy(), k)
    np.save(os.path.join(out_dir, f"merged_{rank}.npy"), merged)
    np.save(os.path.join(out_dir, f"rows_{rank}.npy"), rows)
    dist.barrier()
    dist.destroy_process_group()


def test_topk_allgather_merge_two_ranks(tmp_path):
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    import hop_loader
    hop_loader.load()
    from hop_amd import api
    api.build_library()
    k, world = 32, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, k, str(tmp_path)), nprocs=world, join=True)
    m0 = np.load(tmp_path / "merged_0.npy")
    m1 = np.load(tmp_path / "merged_1.npy")
    assert np.array_equal(m0, m1), "every rank must hold the same merged table"
    rows = np.concatenate([np.load(tmp_path / f"rows_{r}.npy") for r in range(world)])
    ids = rows[:, 1].copy().view(np.int32)
    valid = rows[ids >= 0]
    vid = ids[ids >= 0]
    order = np.lexsort((vid, -valid[:, 0]))[:k]
    assert np.array_equal(m0, valid[order])
    pose, score, mid = api.rows_to_hypos(m0)
    assert (np.diff(score) <= 0).all()
    assert set(mid >> 24) == {0, 1}, "both ranks contribute to the global top-k"


def _strong_worker(rank, world, port, k, H, out_dir):
    """bench.py --scaling strong without the kernels: every rank owns the contiguous block [r H / N, (r + 1) H / N) of ONE
    fixed hypothesis set, packs its k best rows (ids offset by the block start) and takes part in the exchange."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import hop_loader
    hop_loader.load()
    from hop_amd import api
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(13)                      # the same set on every rank (seed 13, as the C5 replay set)
    scores = rng.random(H).astype(np.float32)
    poses = rng.normal(size=(H, 16)).astype(np.float32)
    per = (H + world - 1) // world
    h0, h1 = rank * per, min(H, (rank + 1) * per)
    order = np.lexsort((np.arange(h0, h1), -scores[h0:h1]))[:k]
    rows = np.zeros((k, api.TOPK_ROW_FLOATS), np.float32)
    rows[:, 0] = -np.finfo(np.float32).max
    rows[:, 1] = np.array([-1], np.int32).view(np.float32)[0]
    m = len(order)
    rows[:m, 0] = scores[h0:h1][order]
    rows[:m, 1] = (order.astype(np.int32) + h0).view(np.float32)
    rows[:m, 2:] = poses[h0:h1][order]
    t = torch.from_numpy(rows)
    out = torch.empty((world * k, t.shape[1]), dtype=t.dtype)
    dist.all_gather_into_tensor(out, t)
    merged, n = api.topk_merge(out.numpy(), k)
    np.save(os.path.join(out_dir, f"strong_{rank}.npy"), merged)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("H", [1000, 37])
def test_strong_scaling_shards_reproduce_the_global_topk(tmp_path, H):
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    import hop_loader
    hop_loader.load()
    from hop_amd import api
    k, world = 16, 2
    mp.spawn(_strong_worker, args=(world, _free_port(), k, H, str(tmp_path)), nprocs=world, join=True)
    m0, m1 = np.load(tmp_path / "strong_0.npy"), np.load(tmp_path / "strong_1.npy")
    assert np.array_equal(m0, m1)
    rng = np.random.default_rng(13)
    scores = rng.random(H).astype(np.float32)
    poses = rng.normal(size=(H, 16)).astype(np.float32)
    best = np.lexsort((np.arange(H), -scores))[:k]      # what one GPU holding all H hypotheses would report
    pose, score, ids = api.rows_to_hypos(m0)
    assert np.array_equal(ids, best) and np.array_equal(score, scores[best]) and np.array_equal(pose.reshape(-1, 16), poses[best])


def _frames_worker(rank, world, port, n_frames, out_dir):
    """BASELINE configs[3] without the kernels: frame f belongs to rank f mod world (run_real_all.shard); every rank holds the poses of
    its frames and takes part in the ONE gather at the end of the run (run_real_all.gather_frame_poses)."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import hop_loader
    hop = hop_loader.load()
    from hop_amd import run_real_all as rra
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    keys = [("rec_a", i) for i in range(0, n_frames, 2)] + [("rec_b", i) for i in range(1, n_frames, 2)]
    keys.sort()
    rng = np.random.default_rng(5)
    truth = {k: rng.normal(size=(4, 4)).astype(np.float32) for k in keys}      # the same on every rank: what each frame "computes"
    mine = {k: truth[k] for k in keys if k[1] % world == rank}                # run_raw: frame index mod world inside every record
    got = rra.gather_frame_poses(mine, keys, rank, world, comm=None, dist=dist)
    assert sorted(got) == keys
    np.save(os.path.join(out_dir, f"frames_{rank}.npy"), np.stack([got[k] for k in keys]))
    np.save(os.path.join(out_dir, "frames_truth.npy"), np.stack([truth[k] for k in keys]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [7, 2, 1])
def test_c4_frame_poses_gathered_from_two_ranks(tmp_path, n_frames):
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    import hop_loader
    hop_loader.load()
    from hop_amd import api
    api.build_library()
    port = _free_port()
    mp.spawn(_frames_worker, args=(2, port, n_frames, str(tmp_path)), nprocs=2, join=True)
    t = np.load(tmp_path / "frames_truth.npy")
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"frames_{r}.npy"), t), "every rank ends with every frame's pose"
