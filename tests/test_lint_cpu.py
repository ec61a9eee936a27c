"""CPU (compile-only): the ISA lint for the unconfirmed lane-mask hazard family (profiles/experiments/r02_lane_mask.md,
r02_square_branch.md).  No shipped kernel (render, backward, planes, dense) may contain a select on a lane mask that was
combined on the scalar unit from VALU compares (shape S1 of tools/lint_lane_masks.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_scalar_combined_lane_mask_selects_in_shipped_kernels():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lint_lane_masks.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    heads = [l for l in r.stdout.splitlines() if l.startswith("nfe_")]
    assert len(heads) == 4 and all("S1" in l for l in heads), r.stdout[:500]
    for l in heads:
        assert "S1 (select on scalar-combined VALU mask) = 0," in l, l


def test_inline_asm_audit_and_accumulate_reg_kernel_register_contract():
    """tools/asm_audit.py (compile-only): every instruction-bearing `asm` statement of the four .hip files against the hazards hipcc
    does not pad inside asm strings (cdna_hip_programming.md 5.7) - M0 written and consumed inside ONE statement with its wait
    states, the VGPR index mode switched off before the statement ends, no VALU-written SGPR feeding an asm VMEM instruction, no
    MFMA result read by a statement - and the register contract of bwd_accumulate_reg_kernel: no M0 use outside its
    statements, no scratch, 161 VGPRs (the tile v80..v160 is covered), no compiler instruction inside the reserved range."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_audit.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert "asm audit: 0 violations" in r.stdout
    assert "s_set_gpr_idx_on" in r.stdout and "M0-1 (4 >= 4 states)" in r.stdout        # the index-mode statements were seen and checked
    assert "global_load_lds_dwordx4" in r.stdout and "M0-1 (1 >= 1 states)" in r.stdout   # so were conv3_kernel's LDS-DMA statements


def test_asm_audit_detects_the_reproduced_hazard():
    """The detector must see what the hardware punished: the experiment build -DNFE_SOFTPLUS_SCALAR=1 (an inline-asm v_add_f32 that
    reads a v_exp_f32 / v_log_f32 result without the wait state: run-dependent results on MI355X,
    profiles/experiments/r04_asm_trans_hazard.md) has to fail rule TRNS, in every render kernel that contains the statement."""
    env = dict(os.environ, ASM_AUDIT_FLAGS="-DNFE_SOFTPLUS_SCALAR=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_audit.py"), "--files", "nfe_render.hip"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 1, r.stdout[-2000:]
    hits = [l for l in r.stdout.splitlines() if l.startswith("VIOLATION") and "TRNS" in l]
    assert len(hits) >= 100 and all("v_add_f32 reads the result of v_" in l for l in hits), hits[:3]
    kernels = {l.split(":")[0] for l in hits}
    assert any("render_ws_kernel" in k for k in kernels) and any("render_kernel" in k for k in kernels)
