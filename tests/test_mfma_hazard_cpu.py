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
