"""The product's KERNEL SOURCES executed without a GPU: csrc/*.hip compiled unchanged by g++ against the functional HIP model of tests/emu
(64-lane wavefronts in lockstep at every cross-lane operation, workgroups with LDS and __syncthreads, the runtime calls the host side makes)
and driven through the C-ABI by the very `-m gpu` tests that run on the MI355X, in child processes with HOP_TEST_EMU=1.

Why: GPU access to this repository was closed from outside the build from the end of round 3 through round 4.  A kernel that cannot be run
where it is written can still be EXECUTED -- its indices, masks, queues, reductions and operation order checked against the oracle -- just not
timed.  What this is not: a backend (api.lib() loads libhop.so or fails; only these tests set the variable), a statement about gfx950 code
generation, or about the hardware's reciprocal / square-root estimates (modelled as correctly rounded).

This file keeps a selection that finishes in a few minutes in the CPU suite; the whole `-m gpu` suite runs on the model with
    HOP_TEST_EMU=1 python -m pytest tests -m gpu -n 7 --timeout 3000          (hours; log of the round-4 run: profiles/r04_emu_gpu_suite.txt)
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")


@pytest.fixture(scope="module")
def emu_lib():
    r = subprocess.run(["make", "-C", EMU_DIR, "-j4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lib = os.path.join(EMU_DIR, "_build", "libhop_emu.so")
    assert os.path.exists(lib)
    return lib


def _child_pytest(selection, k=None, env=None, timeout=900):
    cmd = [sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-x"] + selection + (["-k", k] if k else [])
    e = dict(os.environ, HOP_TEST_EMU="1")
    e.update(env or {})
    r = subprocess.run(cmd, capture_output=True, text=True, env=e, cwd=ROOT, timeout=timeout)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and " passed" in r.stdout, tail
    return int(r.stdout.strip().splitlines()[-1].split(" passed")[0].split()[-1])


def test_the_model_runs_the_real_library_entry_points(emu_lib):
    """the exported symbols of the model library are the C-ABI's (include/hop.h): the kernels are reached the way a caller reaches them"""
    import ctypes
    import re
    L = ctypes.CDLL(emu_lib)
    text = open(os.path.join(ROOT, "include", "hop.h")).read()
    names = sorted(set(re.findall(r"\b(hop_[a-z0-9_]+)\s*\(", text)) - {"hop_ctx", "hop_comm"})
    missing = [n for n in names if not hasattr(L, n)]
    assert len(names) > 60 and not missing, missing


def test_smoke_on_the_model(emu_lib):
    """__graft_entry__.smoke() -- generator, ICP, computeLCP, signed distance and the collision decisions against the oracle -- with every
    kernel executed by the model"""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import hop_loader; hop_loader.load(); from hop_amd import api;"
            "api.LIB_PATH = %r; api._lib = None; import __graft_entry__ as g; g.smoke()") % (ROOT, os.path.join(ROOT, "oracle"), emu_lib)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0 and "smoke ok" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_generator_kernels_against_the_reference_built_goldens_on_the_model(emu_lib):
    """k_ppf_matrix_sym, k_pairs, k_quad_prep, k_quads / k_quad_fit, k_verify_*, k_emit: the hypothesis multiset of the reference's own OpenGR
    build (tests/golden/s4pcs_case1.npz), all three Verify modes -- and once more with the binned quadrilateral stage (k_quads_hash,
    HOP_QUADS_HASH=1: the reference's IndexedNormalSet structure), which had never run anywhere before"""
    sel = [os.path.join("tests", "test_gpu_parity.py")]
    assert _child_pytest(sel, "generator_matches_reference_golden and case1") == 3
    assert _child_pytest(sel, "generator_matches_reference_golden and case1", env={"HOP_QUADS_HASH": "1"}) == 3
    # ... and with the workgroups and work-items of every launch scheduled in a shuffled order (the queues are filled through atomics: the
    # emitted multiset must not depend on who gets which slot)
    assert _child_pytest(sel, "generator_matches_reference_golden and case1", env={"HOP_QUADS_HASH": "1", "EMU_ORDER": "shuffle:5"}) == 3
    assert _child_pytest(sel, "generator_matches_reference_golden and case1", env={"EMU_ORDER": "reverse"}) == 3


def test_render_and_normals_kernels_on_the_model(emu_lib):
    """csrc/hop_render.hip (rasteriser with near-plane clipping and the workgroup-per-big-triangle path, image composition, ordered and reduced
    score sums) and csrc/hop_normals.hip (integral-image normals incl. the reference's example/depth7.png, moving least squares) against their
    oracles: every test of tests/test_gpu_render.py and tests/test_gpu_normals.py"""
    assert _child_pytest([os.path.join("tests", "test_gpu_render.py"), os.path.join("tests", "test_gpu_normals.py")]) == 12


def test_icp_with_the_references_minimiser_on_the_model(emu_lib):
    """k_icp_fusedq_momm (the moment sums on the matrix cores: round 5) + k_icp_lm7_solve (nn_mode 7, what the mirrors run): refined poses,
    iteration counts and convergence flags equal to the oracle's BIT FOR BIT, hypotheses that do not converge included"""
    sel = [os.path.join("tests", "test_gpu_zy_icp_canon.py")]
    assert _child_pytest(sel, "not_converged", timeout=1500) == 1
    # (the C1 frame's 100 hypotheses -- test_nn_mode7_c1_depth7_bits_and_distance_to_eigens_run -- and the rest of that file: the whole-suite run
    # on the model, profiles/r05_emu_gpu_suite.txt; kept out of the CPU suite for its minute)
    # (the vector-unit kernel, HOP_ICP_MFMA=0: test_the_two_moment_kernels_return_the_same_integers of the selection below)


def test_gfx950_primitives_as_the_model_states_them(emu_lib):
    """tests/test_gpu_zzzz_dev_selftest.py on the model: the stand-ins of tests/emu/hip/hip_runtime.h for v_med3_u32, v_mad_i32_i24, v_cvt_pk_i16_i32,
    v_dot2_i32_i16, v_perm_b32, v_med3_f32 and v_mfma_i32_16x16x64_i8 against the numpy statement of the same instructions, and the two moment
    kernels against each other (on a device the same file checks the instructions themselves)"""
    assert _child_pytest([os.path.join("tests", "test_gpu_zzzz_dev_selftest.py")], timeout=1500) == 6


def test_the_model_sees_a_staging_slot_reused_before_its_stream_was_synchronised(emu_lib, tmp_path):
    """round 4's model ran every hipMemcpyAsync at the call, so the pinned-staging ring of hop_ctx_h2d -- reuse of a slot
    before hipStreamSynchronize -- was invisible to it.  The model now performs a queued copy as LATE as the stream order allows (the worst
    schedule a device may choose); tests/cpp/emu_async_check.cpp shows that it answers "corrupted" to the premature reuse, "ok" to the legal
    sequences (synchronise first; pageable source), and that EMU_ASYNC=0 is the old behaviour.  Every other test of this file ran with it."""
    exe = str(tmp_path / "emu_async_check")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-I" + EMU_DIR, os.path.join(ROOT, "tests", "cpp", "emu_async_check.cpp"),
                        os.path.join(EMU_DIR, "_build", "emu_runtime.o"), "-lpthread", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = subprocess.run([exe], capture_output=True, text=True, env={k: v for k, v in os.environ.items() if k != "EMU_ASYNC"}).stdout.split()
    assert out == ["staged_then_synced", "ok", "reused_before_sync", "corrupted", "pageable_source", "ok"], out
    old = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, EMU_ASYNC="0")).stdout.split()
    assert old == ["staged_then_synced", "ok", "reused_before_sync", "ok", "pageable_source", "ok"], old


def test_the_models_v_perm_b32_is_the_compilers(emu_lib, tmp_path):
    """k_icp_fusedq_momm's ring read-out gathers bytes with v_perm_b32; which operand the selector bytes 0..3 / 4..7 address is the kind of thing
    a model gets wrong together with the kernel.  LLVM folds __builtin_amdgcn_perm on constants: hipcc's own statement of the instruction, read
    from the gfx950 assembly, must equal the model's (tests/emu/hip/hip_runtime.h emu_perm) for the two selectors the kernel uses."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = tmp_path / "p.hip"
    src.write_text("#include <hip/hip_runtime.h>\n__global__ void kperm(unsigned* o) {\n"
                   "  o[0] = __builtin_amdgcn_perm(0x44434241u, 0x14131211u, 0x07050301u);\n"
                   "  o[1] = __builtin_amdgcn_perm(0x44434241u, 0x14131211u, 0x06040200u);\n}\n")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", str(src), "-o", str(tmp_path / "p.s")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = (tmp_path / "p.s").read_text()
    if "v_perm_b32" in asm:
        pytest.skip("this compiler does not fold the builtin")
    prog = tmp_path / "m.cpp"
    prog.write_text('#include "hip/hip_runtime.h"\nint main() { std::printf("0x%08x 0x%08x\\n", __builtin_amdgcn_perm(0x44434241u, 0x14131211u, 0x07050301u), '
                    '__builtin_amdgcn_perm(0x44434241u, 0x14131211u, 0x06040200u)); }\n')
    exe = str(tmp_path / "m")
    r = subprocess.run(["g++", "-std=c++17", "-I" + EMU_DIR, str(prog), os.path.join(EMU_DIR, "_build", "emu_runtime.o"), "-lpthread", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    model = subprocess.run([exe], capture_output=True, text=True).stdout.split()
    assert model == ["0x44421412", "0x43411311"], model
    for lit in model:
        assert lit in asm, (lit, [ln for ln in asm.splitlines() if "v_mov_b32" in ln])


def _emu_env(emu_lib, **extra):
    """children that find the model under the name libhop.so and the file-based stand-in for RCCL (tests/emu/fake_rccl.cpp) under librccl.so"""
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "HOP_FORCE", "HOP_COMM_ID_FILE", "HOP_GATHER")}
    e["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(emu_lib), "as_libhop") + os.pathsep + e.get("LD_LIBRARY_PATH", "")
    e.update(extra)
    return e


def test_cpp_dataset_driver_gathers_with_two_ranks_on_the_model(emu_lib, tmp_path):
    """BASELINE configs[3] ("frames sharded over the GPUs, RCCL gather of per-frame best pose") with MORE THAN ONE RANK: two processes of the C++
    dataset driver (host/app/run_real_all.cpp) share a record whose frames are already done, meet through the id file (run nonce in its
    name), build their communicators, run ONE hop_frames_allgather and rank 0 writes the table -- the product's exchange code end to end,
    the RCCL calls answered by the inter-process stand-in of tests/emu (files in /dev/shm) and device memory by the model.  The collective on
    real hardware has only ever run with one rank; this is the rank arithmetic, padding and numbering it will run with."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import hop_loader
    hop_loader.load()
    from hop_amd import run_real_all as rr
    exe = os.path.join(ROOT, "icra20-hand-object-pose_amd", "lib", "run_real_all")
    if not os.path.exists(exe):
        pytest.skip("run_real_all is not built")
    base = str(tmp_path / "auto_collect")
    rng = np.random.default_rng(23)
    truth = {}
    for record, frames in (("rec_000", (0, 3, 10, 2, 5)), ("rec_001", (7, 1))):          # rank 0 owns 0, 10, 2 ; rank 1 owns 3, 5, 7, 1
        rec = os.path.join(base, "ellipse", record)
        os.makedirs(rec)
        for k in frames:
            open(os.path.join(rec, f"rgb{k}.png"), "wb").write(b"")
            os.makedirs(os.path.join(rec, "predict", str(k)))
            T = np.eye(4, dtype=np.float32)
            T[:3, :] = rng.normal(size=(3, 4)).astype(np.float32)
            truth[(record, k)] = T
            with open(os.path.join(rec, "predict", str(k), "model2scene.txt"), "w") as f:
                for row in T:
                    f.write(" ".join(f"{v:.9g}" for v in row) + "\n")
    cfg_path = os.path.join(ROOT, "icra20-hand-object-pose_amd", "config", "config_autodataset.yaml")
    adir = rr.write_assets_dir(rr.Assets(), str(tmp_path / "assets"))
    procs = [subprocess.Popen([exe, cfg_path, adir, base, "ellipse"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=_emu_env(emu_lib, HOP_GATHER="1", RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_PORT="29517", HOP_RUN_ID="twoRanks",
                                           HOP_COMM_WAIT_S="60")) for r in (1, 0)]          # (rank 1 first: it waits for rank 0's id)
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so + se
    assert "rank 1 of 2: poses of 7 frames gathered (4 from this rank, 4 rows per rank, through hop_frames_allgather)" in outs[0][0], outs[0][0]
    assert "rank 0 of 2: poses of 7 frames gathered (3 from this rank, 4 rows per rank, through hop_frames_allgather)" in outs[1][0], outs[1][0]
    lines = open(os.path.join(base, "ellipse", "model2scene_all.txt")).read().strip().splitlines()
    assert [(ln.split()[0], int(ln.split()[1])) for ln in lines] == sorted(truth)
    for ln in lines:
        t = ln.split()
        assert np.array_equal(np.array(t[2:], np.float32).reshape(4, 4), truth[(t[0], int(t[1]))])
    assert not [n for n in os.listdir(os.path.join(base, "ellipse")) if n.startswith(".hop_comm_id")], "rank 0 removes its id file"


@pytest.mark.parametrize("world", [2, 3])
def test_topk_exchange_with_several_ranks_on_the_model(emu_lib, tmp_path, world):
    """BASELINE configs[4]'s exchange with more than one rank: every rank packs its shard's table on the "device", one all-gather, the merge
    kernel (one workgroup, bitonic sort of the gathered keys in LDS) -- against the host merge (hop_topk_merge) of the same tables: identical
    rows on every rank, in HypoCompare order with the canonical key (equal scores across ranks, -0, a NaN, shards with fewer than k rows,
    an empty shard)."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import hop_loader
    hop_loader.load()
    from hop_amd import api
    rng = np.random.default_rng(world)
    k = 16
    cases = {}
    sizes = [200, 37, 5, 64]
    for c, n in enumerate(sizes):
        poses = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
        poses[:, :3, 3] = rng.normal(size=(n, 3)).astype(np.float32)
        scores = (rng.integers(0, 12, n) / 4.0).astype(np.float32)          # many ties across the shards
        if c == 1:
            scores[3], scores[20], scores[30] = -0.0, 0.0, np.nan
        cases[f"poses{c}"], cases[f"scores{c}"] = poses, scores
    if world == 3:
        cases["poses2"], cases["scores2"] = cases["poses2"][:2], cases["scores2"][:2]     # two hypotheses for three ranks: an empty shard
    cases["n_cases"] = np.array(len(sizes))
    path = str(tmp_path / "cases.npz")
    np.savez(path, **cases)
    child = os.path.join(EMU_DIR, "exchange_rank.py")
    id_file = str(tmp_path / "id.bin")
    procs = [subprocess.Popen([sys.executable, child, str(r), str(world), id_file, str(k), path, emu_lib], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=_emu_env(emu_lib)) for r in reversed(range(world))]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-2000:] + se[-3000:]
    per_rank = [[ln.split() for ln in so.strip().splitlines() if ln.startswith("case")] for so, _ in reversed(outs)]      # rank order
    for c in range(len(sizes)):
        merged = [np.frombuffer(bytes.fromhex(pr[c][3]), np.uint32).reshape(k, -1) for pr in per_rank]
        tables = np.stack([np.frombuffer(bytes.fromhex(pr[c][4]), np.uint32).view(np.float32).reshape(k, -1) for pr in per_rank])
        for m in merged[1:]:
            assert np.array_equal(m, merged[0]), "every rank holds the same merged table"
        host, n = api.topk_merge(tables, k)           # (host function: no device, no model)
        assert int(per_rank[0][c][2]) == n
        assert np.array_equal(host.view(np.uint32).reshape(k, -1)[:n], merged[0][:n]), c
        ids = merged[0][:n, 1].view(np.int32)
        assert len(set(ids.tolist())) == n and ids.min() >= 0
