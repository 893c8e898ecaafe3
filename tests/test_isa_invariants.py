"""Code-generation invariants of the shipped rollout kernels (CPU: hipcc cross-compiles gfx950 to assembly here).

Round 3 found three step-loop stalls that no source line shows: a loop-carried value (the gathered row, the schedule
values fetched one step ahead) that the register allocator copies at the end of the step puts an `s_waitcnt vmcnt(0)`
-- a wait for the row gather AND the table store -- behind the step barrier of every step.  These
tests compile the two rollout translation units to assembly and check, for the builds the BASELINE shapes run, that

* the first instruction behind a step barrier is not a full vector-memory wait,
* nothing spills (ScratchSize 0),
* the turnstile kernel keeps its four workgroups per CU (<= 128 VGPRs).
"""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "dist_classicrl_amd" / "csrc"
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

LANE_KERNELS = {
    "headline, sparse build": "k_rollout_laneIfNS_7HashEnvELi4ELi128ELb0ELi1ELb1ELb1ELb1",
    "headline, sparse build with the delta log": "k_rollout_laneIfNS_7HashEnvELi4ELi128ELb0ELi2ELb1ELb1ELb1",
    "c2, dataflow kernel": "k_rollout_dfIfNS_7HashEnvELi2ELb0ELi1ELb1",
    "dataflow kernel, 16 actions": "k_rollout_dfIfNS_7HashEnvELi4ELb0ELi1ELb1",
    "dataflow kernel, 16 actions, delta log": "k_rollout_dfIfNS_7HashEnvELi4ELb0ELi2ELb1",
}
TURN_KERNELS = {
    "c3": "k_step_turnIfNS_7HashEnvELi4ELb0",
    "c3 learn_vec": "k_step_turnIfNS_7HashEnvELi4ELb1",
    "c4 shard": "k_step_turnIfNS_7HashEnvELi8ELb0",
    "c5": "k_step_turnIfNS_7HashEnvELi16ELb0",
}


def _assembly(unit, tmp_path_factory):
    if not Path(HIPCC).exists():
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / f"{unit}.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-pass-failed",
           "-DQE_INST_T=float", "-DQE_INST_ENV=HashEnv", "-S", "--cuda-device-only", str(CSRC / f"qe_inst_{unit}.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    return out.read_text().split("\n")


@pytest.fixture(scope="module")
def lane_asm(tmp_path_factory):
    return _assembly("lane", tmp_path_factory)


@pytest.fixture(scope="module")
def step_asm(tmp_path_factory):
    return _assembly("step", tmp_path_factory)


def _kernel(lines, pattern):
    start = next(i for i, l in enumerate(lines) if pattern in l.split(":")[0] and ":" in l and not l.startswith(("\t", ".", ";", " ")))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    meta_end = next(i for i in range(end, len(lines)) if "; Occupancy" in lines[i])
    meta = {}
    for l in lines[end:meta_end + 1]:
        m = re.search(r"; (NumVgprs|ScratchSize|Occupancy): (\d+)", l)
        if m:
            meta[m.group(1)] = int(m.group(2))
    return lines[start:end], meta


@pytest.mark.parametrize("name", list(LANE_KERNELS))
def test_no_full_vector_memory_wait_behind_a_step_barrier(lane_asm, name):
    body, meta = _kernel(lane_asm, LANE_KERNELS[name])
    assert meta["ScratchSize"] == 0, meta
    barriers = 0
    for i, l in enumerate(body):
        # the step barriers are inline assembly: ;;#ASMSTART / s_waitcnt ... / s_barrier / ;;#ASMEND
        if l.strip() == "s_barrier" and "#ASMSTART" in body[i - 2]:
            barriers += 1
            j = i + 1
            while not body[j].strip() or body[j].strip().startswith(";"):
                j += 1
            assert not re.match(r"s_waitcnt vmcnt\(0\)", body[j].strip()), (
                f"{name}: `{body[j].strip()}` right behind the barrier at line {i} of the kernel: a loop-carried value is "
                "being copied at the end of the step (see DESIGN section 9)")
    assert barriers >= 1


@pytest.mark.parametrize("name", list(TURN_KERNELS))
def test_turnstile_kernel_keeps_four_workgroups_per_cu(step_asm, name):
    _body, meta = _kernel(step_asm, TURN_KERNELS[name])
    assert meta["ScratchSize"] == 0, meta
    assert meta["NumVgprs"] <= 128, meta
