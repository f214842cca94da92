"""The build-time assembly scans (ddnerf_amd/csrc/check_asm_hazards.py) on synthetic listings: they guard properties nobody else checks
(an unpadded wait state, a register touched while a scalar load the compiler does not know about is still in flight), so they must
themselves be shown to fire."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ddnerf_amd", "csrc"))
import check_asm_hazards as C  # noqa: E402

HEAD = "_Z9my_kernelILb1EEvPKf:\n"
LOAD = "\t;#ASMSTART\n\ts_load_dwordx2 s[20:21], s[4:5], s9\n\t;#ASMEND\n"
WAIT = "\t;#ASMSTART\n\ts_waitcnt lgkmcnt(0)\n\t;#ASMEND\n"
USE = "\t;#ASMSTART\n\tv_cndmask_b32 v3, 0, v3, s[20:21]\n\t;#ASMEND\n"
END = "\ts_endpgm\n"


def _scan(tmp_path, body):
    p = tmp_path / "k.s"
    p.write_text(HEAD + body + END)
    return C.check_hidden_sloads(str(p), "my_kernel")


def test_hidden_scalar_load_scan(tmp_path):
    # the intended shape: request, unrelated work, wait, use
    n, bad = _scan(tmp_path, LOAD + "\tv_mfma_f32_32x32x2_f32 a[0:15], v1, v2, a[0:15]\n\ts_add_u32 s30, s30, 4\n" + WAIT + USE)
    assert (n, bad) == (1, [])
    # a compiler-issued wait (the one in front of a barrier) completes the load as well
    n, bad = _scan(tmp_path, LOAD + "\ts_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier\n" + USE)
    assert (n, bad) == (1, [])
    # a copy of the destination before the wait: moves stale bits
    n, bad = _scan(tmp_path, LOAD + "\ts_mov_b64 s[40:41], s[20:21]\n" + WAIT + USE)
    assert n == 1 and len(bad) == 1 and "s_mov_b64" in bad[0]
    # a spill of one half, a use with a partial wait, an overwrite
    assert len(_scan(tmp_path, LOAD + "\tv_writelane_b32 v255, s21, 3\n" + WAIT)[1]) == 1
    assert len(_scan(tmp_path, LOAD + "\ts_waitcnt lgkmcnt(1)\n" + USE + WAIT)[1]) == 1
    assert len(_scan(tmp_path, LOAD + "\ts_mov_b32 s20, 0\n" + WAIT)[1]) == 1
    # never waited for
    assert len(_scan(tmp_path, LOAD)[1]) == 1
    # a scalar load the COMPILER issued (outside an asm block) is its own business
    assert _scan(tmp_path, "\ts_load_dwordx2 s[20:21], s[4:5], 0x10\n\ts_mov_b32 s22, s20\n") == (0, [])
