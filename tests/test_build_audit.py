"""Host-side checks of the build: the machine-code audit of the kNN kernels (image_matching_amd/mfma_audit.py) rejects
a crafted assembly file with a VALU write of an MFMA's SrcC register too close behind it, on the straight path and on a
taken branch, in both text forms it reads; and the build's staleness stamps see a change of flags."""
import os

from image_matching_amd import build as b
from image_matching_amd import mfma_audit


GOOD_S = """
	.text
_Z6kernelPi:
	v_mfma_i32_16x16x64_i8 v[10:13], v[0:3], v[4:7], v[20:23]
	s_nop 2
	v_mov_b32_e32 v22, v40            ; three wait states behind: the documented minimum
	v_mfma_i32_16x16x64_i8 v[10:13], v[0:3], v[4:7], v[10:13]
	v_mov_b32_e32 v40, v41            ; in-place accumulation: SrcC == vDst is not this hazard
	v_mfma_i32_16x16x64_i8 v[30:33], v[0:3], v[4:7], v[24:27]
	v_cmp_gt_i32_e32 vcc, v24, v25    ; a compare reads, it does not write a VGPR
	ds_read_b128 v[24:27], v50        ; not a VALU write
	s_nop 0
	v_add_u32_e32 v50, v24, v51
	s_endpgm
"""

BAD_STRAIGHT_S = """
_Z6kernelPi:
	v_mfma_i32_16x16x64_i8 v[10:13], v[0:3], v[4:7], v[20:23]
	s_nop 1
	v_ashrrev_i32_e32 v23, 1, v40     ; two wait states behind
	s_endpgm
"""

BAD_BRANCH_S = """
_Z6kernelPi:
	v_mfma_i32_16x16x64_i8 v[10:13], v[0:3], v[4:7], v[20:23]
	s_cbranch_scc1 .LBB0_2
	s_nop 7
	v_mov_b32_e32 v20, v40            ; far enough on the fall-through path
	s_endpgm
.LBB0_2:
	v_mov_b32_e32 v21, v40            ; one wait state behind on the taken path
	s_endpgm
"""

BAD_OBJDUMP = """
0000000000001000 <_Z6kernelPi>:
	v_mfma_i32_16x16x64_i8 v[10:13], v[0:3], v[4:7], v[20:23]  // 000000001000: D3B6000A 04520900
	s_cbranch_scc1 2                                           // 000000001008: BF850002 <_Z6kernelPi+0x14>
	s_nop 7                                                    // 00000000100C: BF800007
	s_endpgm                                                   // 000000001010: BF810000
	v_mov_b32_e32 v21, v40                                     // 000000001014: 7E2A0328
	s_endpgm                                                   // 000000001018: BF810000
"""


def test_audit_accepts_the_documented_distance():
    assert mfma_audit.audit(GOOD_S) == []


def test_audit_rejects_a_write_too_close_on_the_straight_path():
    found = mfma_audit.audit(BAD_STRAIGHT_S)
    assert len(found) == 1 and found[0][2] == 2 and "v_ashrrev_i32_e32 v23" in found[0][3]
    assert mfma_audit.audit(BAD_STRAIGHT_S, min_wait=2, min_wait_first=2) == []


def test_audit_follows_taken_branches_in_both_text_forms():
    found = mfma_audit.audit(BAD_BRANCH_S)
    assert len(found) == 1 and found[0][2] == 1 and "v21" in found[0][3]
    found = mfma_audit.audit(BAD_OBJDUMP)
    assert len(found) == 1 and found[0][2] == 1 and "v21" in found[0][3]


def test_built_knn_object_passes_the_audit():
    """What build() checks, on the object this tree was built from (skipped where there is no object, e.g. a box that
    received only the library)."""
    obj = os.path.join(b.OBJ, "knn.o")
    if not os.path.exists(obj):
        import pytest
        pytest.skip("no csrc/build/knn.o in this tree")
    assert mfma_audit.audit_object(obj) == []
    # the scan does see the kernels' MFMAs: with an absurd distance every re-initialised accumulator register shows
    assert len(mfma_audit.audit_object(obj, 12, 12)) > 50


def test_stamps_see_flags_and_defines(monkeypatch):
    src = [s for s in b.sources() if s.endswith("knn.hip")][0]
    before = b._obj_digest(src)
    monkeypatch.setitem(b.EXTRA_FLAGS, "knn.hip", ["-fno-honor-nans", "-DSOMETHING"])
    assert b._obj_digest(src) != before
    monkeypatch.setitem(b.EXTRA_FLAGS, "knn.hip", ["-fno-honor-nans"])
    assert b._obj_digest(src) == before
    monkeypatch.setenv("IMATCH_BUILD_DEFINES", "-DKNN_TRACE")
    assert b._obj(src) != os.path.join(b.OBJ, "knn.o")      # development objects get names of their own
