"""The inline-asm MFMA blocks of the recurrent kernels (csrc/mfma_rec.h) manage their result / operand hazards by hand; the
"chained" forms leave the covers out and rely on what the compiler puts between the blocks.  tools/mfma_hazard_check.py
verifies that property on the generated gfx950 ISA; this test runs it on every kernel file that uses the blocks."""
import importlib.util, os, shutil, subprocess, sys, textwrap
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "mfma_hazard_check.py")


def _tool():
    spec = importlib.util.spec_from_file_location("mfma_hazard_check", TOOL)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _check_text(tmp_path, body):
    p = tmp_path / "k.s"
    p.write_text("kern:\n" + textwrap.dedent(body))
    m = _tool()
    total = bad = 0
    for name, ins, labels in m.parse(str(p)):
        n, b = m.check_function(name, ins, labels, "k.s")
        total += n; bad += b
    return total, bad


def test_checker_accepts_covered_and_chained_blocks(tmp_path):
    total, bad = _check_text(tmp_path, """\
        s_nop 2
        v_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], a[0:3], v[0:3]
        ds_read_b128 v[8:11], v20
        s_waitcnt lgkmcnt(0)
        s_nop 2
        v_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], a[4:7], v[0:3]
        s_nop 7
        s_nop 0
        v_add_f32_e32 v4, v0, v1
        s_endpgm
    """)
    assert (total, bad) == (2, 0)


def test_checker_flags_early_read_of_a_result(tmp_path):
    total, bad = _check_text(tmp_path, """\
        s_nop 2
        v_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], a[0:3], v[0:3]
        s_nop 3
        v_add_f32_e32 v4, v0, v1
        s_endpgm
    """)
    assert total == 1 and bad == 1


def test_checker_follows_branches(tmp_path):
    total, bad = _check_text(tmp_path, """\
        v_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], a[0:3], v[0:3]
        s_cbranch_scc1 .LBB0_2
        s_nop 7
        s_nop 0
        .LBB0_2:
        v_add_f32_e32 v4, v0, v1
        s_endpgm
    """)
    assert total == 1 and bad == 1          # the taken path reaches the read after one wait state


def test_checker_flags_valu_write_of_an_operand_right_before(tmp_path):
    total, bad = _check_text(tmp_path, """\
        v_mov_b32_e32 v8, v30
        v_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], a[0:3], v[0:3]
        s_nop 7
        s_nop 0
        s_endpgm
    """)
    assert total == 1 and bad == 1


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_generated_isa_of_the_recurrent_kernels_has_no_mfma_hazard():
    r = subprocess.run([sys.executable, TOOL], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "no hazard" in r.stdout


def test_drain_checker_recognises_the_back_edge_copy_chain():
    """tools/vmcnt_drain_check.py on a hand-written listing: the round-4 pattern (loop-carried copies of load destinations in front
    of the back-edge branch, wait counts falling to 0) is reported, ordinary waits in front of uses are not"""
    spec = importlib.util.spec_from_file_location("vmcnt_drain_check", os.path.join(ROOT, "tools", "vmcnt_drain_check.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    drained = ["_Zk:", ".LBB0_1:", "s_barrier", "s_waitcnt vmcnt(3)", "v_mov_b32_e32 v59, v30", "s_waitcnt vmcnt(2)", "v_mov_b32_e32 v60, v31",
               "s_waitcnt vmcnt(1)", "v_mov_b32_e32 v58, v32", "s_waitcnt vmcnt(0)", "v_mov_b32_e32 v29, v33", "s_cbranch_scc1 .LBB0_1"]
    # the same run in front of a FORWARD branch (r5: result registers of an exchange poll zeroed in front of its `dead` test) is not one
    forward = ["_Zk:", "s_waitcnt vmcnt(2)", "v_mov_b64_e32 v[16:17], 0", "s_waitcnt vmcnt(1)", "v_mov_b64_e32 v[14:15], 0",
               "s_waitcnt vmcnt(0)", "v_mov_b64_e32 v[12:13], 0", "s_cbranch_vccnz .LBB0_9", "s_nop 0", ".LBB0_9:"]
    fine = ["_Zk:", "s_waitcnt vmcnt(14)", "v_add_f32_e32 v13, v60, v13", "s_waitcnt vmcnt(13)", "v_add_f32_e32 v10, v58, v10",
            "s_waitcnt vmcnt(3)", "v_mov_b32_e32 v1, v2", "s_waitcnt vmcnt(3)", "v_mov_b32_e32 v3, v4", "s_waitcnt vmcnt(3)", "v_mov_b32_e32 v5, v6"]
    assert m.runs(drained) == [("_Zk", 4, 4)] and m.runs(fine) == [] and m.runs(forward) == []


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_generated_isa_of_the_recurrent_kernels_does_not_drain_its_prefetch_rings():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "vmcnt_drain_check.py")], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_the_clobber_free_store_stays_out_of_the_exchange_paths():
    """ADVICE r5 (csrc/common.h): st_su is an asm store the compiler neither sees nor orders - valid for write-only outputs that
    the kernel never re-reads or signals.  It is used by the encoder LSTM's saved tensors only; no kernel that exchanges data
    between workgroups (cluster_xchg.h includers, the decode kernels) may contain it."""
    import glob
    import re
    csrc = os.path.join(ROOT, "self-attention-tacotron_amd", "csrc")
    users = []
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))):
        src = open(f).read()
        code = re.sub(r"//[^\n]*", "", src)
        if re.search(r"\bst_su\s*\(", code) and os.path.basename(f) != "common.h":
            users.append(os.path.basename(f))
            assert "cluster_xchg.h" not in src and "gput" not in code, f
    assert users == ["lstm.hip"], users
