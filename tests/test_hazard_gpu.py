"""GPU: the operand forms the build's assembly pass PRODUCES must be safe on the GPU the suite runs on.
csrc/pk_opsel_fix.py commutes `v_pk_*_f32 ... op_sel:[0,1]` (reads zeros in lanes 48-63 beside an MFMA wave on MI355X,
profiles/experiments/r04_pk_opsel_hazard.md) into `op_sel:[1,0]`; tools/microbench/pk_opsel.hip is the stand-alone reproducer."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("pk") / "pk_opsel_pad")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-DMFMA_PAD", "-o", exe,
                           os.path.join(ROOT, "tools", "microbench", "pk_opsel.hip")], stderr=subprocess.DEVNULL)
    return exe


def _run(exe, form, launches=10):
    out = subprocess.run([exe, str(form), "8", "8", str(launches)], capture_output=True, text=True, timeout=120).stdout
    m = re.search(r": (\d+) wrong of ([0-9.e+]+) lane-results", out)
    assert m, out
    lanes = [int(x.split(":")[0]) for x in out.split("by lane:")[1].split()]
    return int(m.group(1)), float(m.group(2)), lanes


@pytest.mark.parametrize("form", [2, 10, 13, 4, 8])
def test_operand_forms_the_library_contains_never_misread(probe, form):
    """form 2: the commuted encoding the pass writes; 10: `op_sel:[1,0,0]`, the library's commonest; 13: both sources the same pair
    (render_kernel's tap weight); 4: default op_sel; 8: `op_sel_hi:[1,0]` alone, the mirror image of the faulty form (HIGH result
    from src0's high and src1's LOW register), which the pass leaves in place - 8e9 lane results each beside MFMA waves, none wrong."""
    wrong, total, _ = _run(probe, form)
    assert total > 5e9 and wrong == 0, (form, wrong, total)


def test_the_hazardous_form_is_characterised(probe):
    """Informational: on MI355X form 1 (`op_sel:[0,1]`, two different pairs) misreads ~1e-5 of its lane results, always lanes 48-63.
    Hardware without the fault passes too; what must hold is that a fault, if present, has the signature the pass was built for."""
    wrong, total, lanes = _run(probe, 1)
    print(f"v_pk_mul_f32 op_sel:[0,1]: {wrong} wrong of {total:.3g} lane results, lanes {sorted(set(lanes))}")
    assert all(48 <= l <= 63 for l in lanes), lanes
