"""Generated-code check (no GPU): no LDS access in flight at an s_barrier in any kernel of csrc/ (tools/barrier_audit.py).  The hand-rolled
LDS-DMA rings use __builtin_amdgcn_s_barrier(), which is no compiler fence: attn_qkv_fwd_bf16_kernel raced on it until round 4."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import barrier_audit


def test_the_scanner_sees_a_wait_sunk_below_the_barrier(tmp_path):
    racy = """_ZN4gm3d4demoEv:
	ds_read_b128 v[40:43], v30 offset:16384
	s_waitcnt vmcnt(0)
	s_barrier
	s_waitcnt lgkmcnt(0)
	v_mfma_f32_32x32x16_bf16 a[0:15], v[40:43], v[36:39], a[0:15]
.LBB0_1:
	ds_read_b128 v[40:43], v30
	s_waitcnt vmcnt(0) lgkmcnt(0)
	s_barrier
	s_endpgm
"""
    f = tmp_path / "demo.s"
    f.write_text(racy)
    assert barrier_audit.findings(str(f)) == [("_ZN4gm3d4demoEv", 4, 1)]


def test_no_kernel_reaches_a_barrier_with_lds_accesses_in_flight():
    assert barrier_audit.audit() == []
