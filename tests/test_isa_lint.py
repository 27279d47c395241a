"""The build-time ISA lint (tools/isa_defuse_lint.py): vector registers read but never written.

hipcc miscompiled one triplet-attention backward instantiation exactly this way (a spilled 128-bit
MFMA operand came back one dword short), so `_lib.build_library()` lints every kernel it ships; these
cases pin the lint itself on small hand-written listings."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tools'))
import isa_defuse_lint as lint     # noqa: E402

GOOD = """
	.text
_Z4goodv:                               ; @_Z4goodv
; %bb.0:
	v_lshlrev_b32_e32 v1, 2, v0
	ds_read_b128 v[2:5], v1
	v_accvgpr_write_b32 a3, v1              ;  Reload Reuse
	buffer_load_dwordx4 a[4:7], v1, s[0:3], 0 offen
	scratch_load_dwordx3 a[8:10], off, off ; 12-byte Folded Reload
	v_accvgpr_mov_b32 a11, a3
.LBB0_1:
	v_mfma_f32_32x32x16_bf16 a[16:31], v[2:5], a[8:11], 0
	v_mfma_f32_32x32x16_bf16 a[16:31], v[2:5], a[4:7], a[16:31]
	v_accvgpr_read_b32 v6, a16
	global_store_dword v1, v6, s[4:5]
	s_endpgm
.Lfunc_end0:
"""

# the shape of the real miscompile: a[8:10] reloaded from scratch, the 4th dword parked in a3 and forgotten
BAD = GOOD.replace('\tv_accvgpr_mov_b32 a11, a3\n', '')

PACKED = """
_Z6packedv:
	v_mov_b32_e32 v4, 1.0
	v_mov_b32_e32 v6, 2.0
	v_mov_b32_e32 v7, 3.0
	v_pk_mul_f32 v[8:9], v[4:5], v[6:7] op_sel_hi:[0,1]
	global_store_dwordx2 v0, v[8:9], s[0:1]
	s_endpgm
.Lfunc_end1:
"""


def test_defined_registers_pass():
    assert lint.lint_text(GOOD) == {}


def test_unrestored_spill_dword_is_flagged():
    assert lint.lint_text(BAD) == {'_Z4goodv': ['a11']}


def test_vgprs_are_checked_too():
    # (v0..v2 arrive initialised; everything above must be written by someone)
    assert lint.lint_text(GOOD.replace('ds_read_b128 v[2:5], v1', 'ds_read_b64 v[2:3], v1')) == {'_Z4goodv': ['v4', 'v5']}


def test_packed_broadcast_reads_one_half():
    assert lint.lint_text(PACKED) == {}                                      # v5 is not selected
    assert lint.lint_text(PACKED.replace(' op_sel_hi:[0,1]', '')) == {'_Z6packedv': ['v5']}


def test_kernels_are_separate():
    both = lint.lint_text(GOOD + BAD.replace('_Z4goodv', '_Z3badv'))
    assert both == {'_Z3badv': ['a11']}


# the store-data hazard met on MI355X (csrc/triplet_attention16.hip, fp16 column-sum variant): a 128-bit store whose first
# data register a VALU instruction rewrites in the very next slot
STORE_HAZARD = """
_Z5storev:
	ds_read_b128 v[114:117], v48
	s_waitcnt lgkmcnt(0)
	buffer_store_dwordx4 v[114:117], v72, s[12:15], s28 offen
	v_cvt_f32_f16_e32 v114, v108
	s_endpgm
.Lfunc_end2:
"""


def test_store_data_overwritten_behind_a_wide_store_is_flagged():
    bad = lint.lint_store_hazard(STORE_HAZARD)
    assert list(bad) == ['_Z5storev'] and 'v114' in bad['_Z5storev'][0]
    assert '_Z5storev' in lint.lint_all(STORE_HAZARD)
    padded = STORE_HAZARD.replace('\tv_cvt_f32_f16_e32 v114, v108\n', '\ts_nop 0\n\tv_cvt_f32_f16_e32 v114, v108\n')
    assert lint.lint_store_hazard(padded) == {}
    other = STORE_HAZARD.replace('v_cvt_f32_f16_e32 v114, v108', 'v_cvt_f32_f16_e32 v120, v108')      # not a data register
    assert lint.lint_store_hazard(other) == {}
    narrow = STORE_HAZARD.replace('buffer_store_dwordx4 v[114:117]', 'buffer_store_dwordx2 v[114:115]')  # 64-bit stores are safe
    assert lint.lint_store_hazard(narrow) == {}
