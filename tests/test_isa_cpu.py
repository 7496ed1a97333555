"""What the gfx950 assembly of the hot kernels must keep (hipcc cross-compiles without a GPU): the properties DESIGN section 4 quotes are
checked in the machine code, not in the source -- a compiler that sinks a load behind a branch again, or a source change that spills
the kernels that have run on hardware, fails here.  One compilation of csrc/hop_kernels.hip (~1 min)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "icra20-hand-object-pose_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    d = tmp_path_factory.mktemp("isa")
    kflags = re.search(r"^KERNELS_FLAGS := (.*)$", open(os.path.join(SRC, "Makefile")).read(), re.M).group(1).split()   # the unit's own flags
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", *kflags, "-c", os.path.join(SRC, "hop_kernels.hip"),
                        "-o", str(d / "k.o"), "--save-temps", "-Rpass-analysis=kernel-resource-usage"], cwd=str(d), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    usage, cur = {}, None
    for ln in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = usage.setdefault(m.group(1), {})
            continue
        m = re.search(r":\s+([A-Za-z][A-Za-z /\[\]]*): (\d+) \[-Rpass", ln)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    asm = open(d / "hop_kernels-hip-amdgcn-amd-amdhsa-gfx950.s").read()

    def body(mangled):
        i = asm.index(mangled + ":")
        return [ln.strip() for ln in asm[i:asm.index(".Lfunc_end", i)].splitlines() if ln.strip() and not ln.strip().startswith((";", "."))]
    yield usage, body
    shutil.rmtree(d, ignore_errors=True)


MOMM = "_ZN3hop17k_icp_fusedq_mommENS_7IcpArgsEi"
LCP = "_ZN3hop16k_lcp_cells_fastILb1EEEvNS_7LcpArgsEiiii"


def test_registers_scratch_and_occupancy_of_the_icp_kernels(isa):
    usage, _ = isa
    want = {MOMM: (80, 6),                                            # the shipped nn_mode 7 kernel: 6 waves per SIMD, nothing in scratch
            "_ZN3hop12k_icp_fusedqILb1EEEvNS_7IcpArgsEi": (72, 7),    # nn_mode 4 as it ran on hardware in round 2
            "_ZN3hop16k_icp_fusedq_momENS_7IcpArgsEi": (120, 4),      # nn_mode 6 as it ran in round 3
            "_ZN3hop17k_icp_fusedq_momiENS_7IcpArgsEi": (152, 3),     # nn_mode 7 on the vector units
            LCP: (64, 7)}
    for k, (vg, occ) in want.items():
        u = usage[k]
        assert u["ScratchSize [bytes/lane]"] == 0, (k, u)
        assert u["VGPRs"] <= vg and u["Occupancy [waves/SIMD]"] >= occ, (k, u)


def test_a_tested_record_is_one_gather(isa):
    _, body = isa
    m = body(MOMM)
    # the 8-byte cell record of the packed lookup is ONE dwordx2 gather in both instantiations of the lookup (first pass, dense deferred pass);
    # round 5's assembly had none: .y at +4, a branch, then .x.  (The +4 loads that remain belong to q_chunk_exact, the rare exact re-scan.)
    assert sum(bool(re.match(r"global_load_dwordx2 v\[\d+:\d+\], v\[\d+:\d+\], off$", x)) for x in m) >= 2
    lcp = body(LCP)
    # computeLCP's two 16-byte head records: whole dwordx4 gathers, and no lone load of their .w (+12) followed by the xyz part
    assert sum(bool(re.match(r"global_load_dwordx4 v\[\d+:\d+\], v\[\d+:\d+\], off$", x)) for x in lcp) >= 2
    for i, x in enumerate(lcp):
        if re.match(r"global_load_dword v\d+, v\[(\d+:\d+)\], off offset:12$", x):
            regs = re.match(r"global_load_dword v\d+, v\[(\d+:\d+)\], off offset:12$", x).group(1)
            assert not any(re.match(r"global_load_dwordx3 v\[\d+:\d+\], v\[%s\], off$" % regs, y) for y in lcp[i + 1:i + 12]), (x, lcp[i:i + 12])


def test_the_poses_stay_scalar_in_compute_lcp_and_leave_the_tag_pipe_in_the_icp_kernel(isa):
    _, body = isa
    lcp = body(LCP)
    assert sum(x.startswith("s_load_dword") for x in lcp) >= 20                      # poses, inverse poses, list geometry: scalar loads
    assert not any(re.match(r"global_load_dwordx4 v\[\d+:\d+\], v\d+, s\[", x) for x in lcp)   # no uniform-address VECTOR load of them
    m = body(MOMM)
    n_uniform = sum(bool(re.match(r"global_load_dwordx4 v\[\d+:\d+\], v\d+, s\[\d+:\d+\]( offset:\d+)?$", x)) for x in m)
    # what remains with a scalar base are the chunk pairs of the list scan (2 per scan loop, 3 scan loops in the text); the 45 uniform
    # re-loads of pose / inverse pose / accumulated transform are gone (staged in LDS, read by broadcast)
    assert n_uniform <= 8, n_uniform
    assert sum(x.startswith("v_mfma_i32_16x16x64_i8") for x in m) == 9


def test_packed_f32_is_off_in_the_two_shipped_lookup_kernels_and_nowhere_else(isa):
    """The unit is compiled with packed-fp32-ops off and every kernel but k_icp_fusedq_momm / k_lcp_cells_fast switches it back on (HOP_PK_F32): a
    v_pk_*_f32 pair issues no faster than its two scalar halves on this part and costs v_movs to set up (profiles/r02_valu_issue_rates.txt).  Nothing
    may have become a CALL by it (a kernel whose target features differ from the HIP header's stops inlining __syncthreads / __ballot)."""
    _, body = isa
    pk = re.compile(r"v_pk_(fma|mul|add)_f32")
    for k in (MOMM, LCP, "_ZN3hop16k_lcp_cells_fastILb0EEEvNS_7LcpArgsEiiii"):
        b = body(k)
        assert not any(pk.match(x) for x in b), k
        assert not any(x.startswith("s_swappc") for x in b), k
    for k in ("_ZN3hop12k_icp_fusedqILb1EEEvNS_7IcpArgsEi", "_ZN3hop16k_icp_fusedq_momENS_7IcpArgsEi", "_ZN3hop7k_quadsENS_8QuadArgsEi"):
        b = body(k)
        assert any(pk.match(x) for x in b), k     # (as they ran on hardware)
        assert not any(x.startswith("s_swappc") for x in b), k
