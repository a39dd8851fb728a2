"""CPU: the build-time check of the loads the compiler does not see (tools/check_hidden_loads.py).

The paired-end read loop at seven and eight isoforms (kernels_grp.inl pe_dense) and the single-end descriptor loop
(kernels_flat.inl flat_units_desc) fetch ahead with inline-assembly `global_load`s and wait with a hand-written
`s_waitcnt`.  The checker is exercised on made-up assembly (a clean trip, a register touched before the wait, a wait
count larger than the operations behind the load) and, when the build left its assembly behind
(miso_amd/csrc/.isa/*.s), on the real thing -- the Makefile runs the same check as part of `all`."""
import glob
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_hidden_loads", os.path.join(ROOT, "tools", "check_hidden_loads.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)

TRIP = """
.LBB0_1:
	;;#ASMSTART
	global_load_dwordx4 v[10:13], v[2:3], off
	;;#ASMEND
	v_add_co_u32_e32 v2, vcc, 16, v2
	;;#ASMSTART
	global_load_dwordx2 v[14:15], v[2:3], off
	;;#ASMEND
	v_mul_f64 v[20:21], v[22:23], v[24:25]
{body}
	;;#ASMSTART
	s_waitcnt vmcnt({n})
	;;#ASMEND
	v_mov_b32_e32 v30, v10
	s_cbranch_scc1 .LBB0_1
	s_endpgm
"""
GATHERS = "\n".join("\tglobal_load_dword v%d, v[40:41], off" % (50 + i) for i in range(4))


def _check(tmp_path, body, n):
    f = tmp_path / "k.s"
    f.write_text(TRIP.format(body=body, n=n))
    return chk.check_file(str(f))


def test_a_clean_trip_passes(tmp_path):
    n, problems = _check(tmp_path, GATHERS, 4)
    assert n == 1 and problems == []


def test_a_register_of_the_load_touched_before_the_wait_is_reported(tmp_path):
    for bad in ("\tv_mov_b32_e32 v31, v12", "\tscratch_store_dword off, v15, s0", "\tv_add_u32_e32 v11, v1, v2"):
        n, problems = _check(tmp_path, GATHERS + "\n" + bad, 4)
        assert n == 1 and len(problems) == 1 and "touches" in problems[0], (bad, problems)


def test_a_wait_count_larger_than_the_operations_behind_the_load_is_reported(tmp_path):
    three = "\n".join(GATHERS.split("\n")[:3])
    n, problems = _check(tmp_path, three, 4)
    assert n == 1 and len(problems) == 1 and "only 3 younger" in problems[0], problems
    # operations under a forward branch (the cold path) are not on every path: they do not count
    cold = three + "\n\ts_cbranch_vccz .LBB0_9\n\tglobal_load_dword v60, v[40:41], off\n.LBB0_9:"
    n, problems = _check(tmp_path, cold, 4)
    assert len(problems) == 1 and "only 3 younger" in problems[0], problems
    # vmcnt(0) needs none
    n, problems = _check(tmp_path, "", 0)
    assert problems == []


def test_the_build_s_own_assembly():
    files = sorted(glob.glob(os.path.join(ROOT, "miso_amd", "csrc", ".isa", "*.s")))
    if not files:
        import pytest
        pytest.skip("no assembly kept by the build here (miso_amd/csrc/.isa: the Makefile writes it where it compiles)")
    groups = 0
    for f in files:
        n, problems = chk.check_file(f)
        assert problems == [], problems[:5]
        groups += n
    assert groups > 0
