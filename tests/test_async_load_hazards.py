"""No compiled kernel may touch a register that a hand-issued asynchronous load has not delivered yet.

The N-split kernels request their weight blocks in `asm volatile` statements and wait for them with hand-counted
`s_waitcnt vmcnt(N)` (savad_kernels.h: wload_frag / wwait); the compiler takes the destination registers for valid as soon as
the statement has executed and may copy or reuse them while the load is in flight.  scripts/check_async_loads.py walks the
control-flow graph of every kernel in the gfx950 assembly of csrc/savad.hip with the queue of loads in flight and reports such
accesses.  Round 4 found three kernels with them (a block requested and never waited for, registers reused; four registers of a
requested block parked in AGPRs before the wait) behind one wrong 32-row tile in a few percent of the runs of a multi-stream GPU
test; this test keeps them out.  CPU only: it disassembles the device code of the library `build()` made (the checker also takes compiler
assembly: `python scripts/check_async_loads.py`; both forms find 531 / 56 / 464 accesses in the three kernels as they were)."""
import importlib.util
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent


def _checker():
    spec = importlib.util.spec_from_file_location("check_async_loads", REPO / "scripts" / "check_async_loads.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


ASM_RACE = """
_Z4demoPf:                              ; @_Z4demoPf
; %bb.0:
	;;#ASMSTART
	global_load_dwordx4 v[4:7], v1, s[2:3] offset:0
	;;#ASMEND
	;;#ASMSTART
	global_load_dwordx4 v[8:11], v1, s[2:3] offset:1024
	;;#ASMEND
	s_cbranch_scc0 .LBB0_2
; %bb.1:
	v_accvgpr_write_b32 a0, v4
.LBB0_2:
	;;#ASMSTART
	s_waitcnt vmcnt(1)
	;;#ASMEND
	v_add_f32_e32 v0, v4, v5
	{tail}
	s_endpgm
.Lfunc_end0:
"""


def test_the_checker_sees_copies_reuse_and_respects_waits(tmp_path):
    chk = _checker()
    # (a) a copy of a requested register in one arm of a branch; (b) the younger block used behind vmcnt(1): still in flight
    f = tmp_path / "race.s"
    f.write_text(ASM_RACE.format(tail="v_mov_b32_e32 v8, 0"))
    (sym, (hazards, truncated)), = chk.check_file(f).items()
    assert sym == "_Z4demoPf" and not truncated
    assert {(h[1].split()[0], h[3].split()[1]) for h in hazards} == {("v_accvgpr_write_b32", "v[4:7],"), ("v_mov_b32_e32", "v[8:11],")}
    # the older block behind vmcnt(1) is fine, and so is the younger one behind vmcnt(0)
    f.write_text(ASM_RACE.replace("	v_accvgpr_write_b32 a0, v4\n", "").format(tail="s_waitcnt vmcnt(0)\n\tv_mov_b32_e32 v8, 0"))
    (_, (hazards, _)), = chk.check_file(f).items()
    assert hazards == []


def test_a_branch_taken_before_the_request_carries_no_request(tmp_path):
    """the taken edge of a conditional branch leaves with the queue as it is at the branch (blocks are not split behind branches)"""
    chk = _checker()
    f = tmp_path / "early.s"
    f.write_text("""
_Z5earlyPf:                             ; @_Z5earlyPf
	s_cbranch_scc1 .LBB1_2
	;;#ASMSTART
	global_load_dwordx4 v[4:7], v1, s[2:3] offset:0
	;;#ASMEND
	s_waitcnt vmcnt(0)
.LBB1_2:
	v_mov_b32_e32 v4, 0
	s_endpgm
.Lfunc_end1:
""")
    (_, (hazards, _)), = chk.check_file(f).items()
    assert hazards == []


def test_no_kernel_of_the_built_library_touches_a_register_with_a_load_in_flight():
    """the device code INSIDE voice_activity_detection_amd/libsavad.so (the artifact that ships), every vector-memory load of
    every kernel -- the compiler's own included: it has to wait before it reuses a destination, and does"""
    if not Path("/opt/rocm/lib/llvm/bin/llvm-objdump").exists():
        pytest.skip("llvm-objdump not available")
    from voice_activity_detection_amd.build import build

    chk = _checker()
    report = chk.check_library(build())
    # the kernels that use hand-issued loads must be among the ones checked (an extraction that finds nothing must not pass)
    names = " ".join(report)
    for needle in ("10row_kernelILb0", "10row_kernelILb1", "21packed_forward_kernel", "16input_qkv_kernelE", "attention_pw_kernel_bf16",
                   "row_kernel_bf16ILb0", "input_qkv_kernel_bf16", "12row_kernel_mILb1", "20attention_row_kernelILb0"):
        assert needle in names, f"{needle} was not found in the library's device code"
    assert len(report) >= 40
    bad = {sym: (h[:4], t) for sym, (h, t) in report.items() if h or t}
    assert not bad, f"registers touched while their load is in flight: {bad}"


ASM_LDS = """
_Z4ringPf:                              ; @_Z4ringPf
; %bb.0:
	ds_read_b128 v[20:23], v2 offset:64
	;;#ASMSTART
	ds_read_b128 v[4:7], v1 offset:0
	;;#ASMEND
	;;#ASMSTART
	ds_read_b128 v[8:11], v1 offset:1024
	;;#ASMEND
	s_load_dwordx2 s[4:5], s[0:1], 0x0
	;;#ASMSTART
	s_waitcnt lgkmcnt({n})
	;;#ASMEND
	v_mfma_f32_32x32x16_bf16 v[32:47], v[4:7], v[12:15], v[32:47]
	s_endpgm
.Lfunc_end0:
"""


def test_hand_issued_lds_reads_are_tracked(tmp_path):
    """round 5: ds_read_b128 in asm statements against counted lgkmcnt waits (the bf16 ring GEMMs).  One read may stay in flight
    behind lgkmcnt(1) -- the younger one; with lgkmcnt(2) the MFMA reads a fragment that has not landed.  The scalar load in
    between is not part of the queue (it can only make the hardware's wait retire more)."""
    chk = _checker()
    f = tmp_path / "lds.s"
    f.write_text(ASM_LDS.format(n=1))
    (_, (hazards, _)), = chk.check_file(f).items()
    assert hazards == []
    f.write_text(ASM_LDS.format(n=2))
    (_, (hazards, _)), = chk.check_file(f).items()
    assert len(hazards) == 1 and hazards[0][1].startswith("v_mfma") and hazards[0][3].startswith("ds_read_b128 v[4:7]")


ASM_DMA = """
_Z4ringPf:                              ; @_Z4ringPf
; %bb.0:
	global_load_lds_dwordx4 v1, s[2:3]
.LBB0_1:
	s_waitcnt vmcnt(0)
	{barrier}
	global_load_lds_dwordx4 v1, s[2:3] offset:1024
	ds_read_b128 v[4:7], v2
	s_waitcnt lgkmcnt(0)
	v_mfma_f32_32x32x16_bf16 v[32:47], v[4:7], v[12:15], v[32:47]
	s_cbranch_scc1 .LBB0_1
	s_endpgm
.Lfunc_end0:
"""


def test_a_ring_without_its_barrier_is_reported(tmp_path):
    """round 5: LDS-DMA destinations.  barrier -> request -> read -> barrier -> request is the protocol; without the barrier the
    second request follows the wave's own LDS read directly, and may land on a block other waves still read."""
    chk = _checker()
    f = tmp_path / "dma.s"
    f.write_text(ASM_DMA.format(barrier="s_barrier"))
    assert all(h == [] for h, _ in chk.check_file(f).values())
    f.write_text(ASM_DMA.format(barrier="s_nop 0"))
    (_, (hazards, _)), = chk.check_file(f).items()
    assert len(hazards) == 1 and hazards[0][1].startswith("global_load_lds") and hazards[0][3].startswith("ds_read_b128")


@pytest.fixture(scope="module")
def fault_asm(tmp_path_factory):
    """csrc/savad.hip compiled to assembly with every fault of SAVAD_FAULT_INJECT switched on (savad_kernels.h)"""
    import subprocess

    from voice_activity_detection_amd.build import hipcc

    out = tmp_path_factory.mktemp("fault") / "fault23.s"
    subprocess.run([hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "--cuda-device-only", "-S", "-DSAVAD_FAULT_INJECT=23",
                    str(REPO / "voice_activity_detection_amd" / "csrc" / "savad.hip"), "-o", str(out)], check=True)
    return out


def test_deliberately_broken_builds_fail_the_check(fault_asm):
    """NEGATIVE test on the real sources: one wwait removed from the single-launch fp32 forward (bit 1), the bf16 ring GEMMs waiting
    for one LDS fragment too few (bit 2), the bf16 weight ring without its workgroup barrier (bit 4), ring block 2 of the bf16 row chain
    handed over at its barrier without the counted wait (bit 16; bits 1 and 16 are what the GPU suite's negative test runs).  Each must show up, in the
    kernel it was planted in and as the kind of hazard it is."""
    chk = _checker()
    report = chk.check_file(fault_asm)
    by = lambda key: [h for sym, (hz, _) in report.items() if key in sym for h in hz]
    fp32 = by("21packed_forward_kernelE")
    assert fp32 and all(h[3].startswith("global_load_dwordx4") for h in fp32)             # registers of a block still in flight
    row = by("15row_kernel_bf16ILb0ELi4E")
    assert any(h[1].startswith("v_mfma") and h[3].startswith("ds_read_b128") for h in row)   # a fragment used before it landed
    assert any(h[1].startswith("global_load_lds") for h in row)                              # a ring slot re-targeted without the barrier
    assert any(h[1].startswith("global_load_lds") and "published before it has landed" in h[3] for h in row)   # bit 16: block 2 handed over unwaited
    packed = by("26packed_forward_kernel_bf16ILi4ELi2ELi0E")
    assert any(h[1].startswith("v_mfma") and h[3].startswith("ds_read_b128") for h in packed) and any(h[1].startswith("global_load_lds") for h in packed)


def test_a_ring_wait_that_counts_one_store_too_many_is_reported(tmp_path_factory):
    """NEGATIVE test of the publication rule on the real sources (bit 8 of SAVAD_FAULT_INJECT, on its own: bit 4 removes the very
    barriers the rule counts at): the ring waits in front of the bf16 row launch's Q / K / V steps leave the wave's eight stores in
    flight -- vmcnt(8).  With vmcnt(9) the newest DMA piece of the block being waited for may still be in flight when the wave
    meets the others at the barrier; the product build has no such DMA (test_no_kernel_of_the_built_library_...)."""
    import subprocess

    from voice_activity_detection_amd.build import hipcc

    out = tmp_path_factory.mktemp("fault8") / "fault8.s"
    subprocess.run([hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "--cuda-device-only", "-S", "-DSAVAD_FAULT_INJECT=8",
                    str(REPO / "voice_activity_detection_amd" / "csrc" / "savad.hip"), "-o", str(out)], check=True)
    report = _checker().check_file(out)
    row = [h for sym, (hz, _) in report.items() if "15row_kernel_bf16ILb0ELi4E" in sym for h in hz]
    assert row and all(h[1].startswith("global_load_lds") and "published before it has landed" in h[3] for h in row)
    assert not [sym for sym, (hz, _) in report.items() if hz and "15row_kernel_bf16ILb0ELi4E" not in sym]   # ... and nothing else moved
