"""CPU, build container only: the reference's kernel file as ROCm's own OpenCL tool chain builds it for gfx950
(oracle/_ref_gfx950/, real ROCm builtin library) — which arithmetic does it contain, and is that the IEEE-plain
contract the oracle is pinned to (oracle/ref_shim.cpp)?  It is not: this test records the fact (DESIGN.md §3,
profiles/r02_ref_gfx950_builtins.md, profiles/r02_ref_distance.json measure the consequences on the GPU)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HSACO = os.path.join(ROOT, "oracle", "_ref_gfx950", "ref950.hsaco")
HSACO_NC = os.path.join(ROOT, "oracle", "_ref_gfx950", "ref950_nocontract.hsaco")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree exists in the build container only")


def ops(path, symbol):
    dis = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", "--disassemble-symbols=" + symbol, path], capture_output=True,
                         text=True, check=True).stdout
    return [re.sub(r"_(e32|e64)$", "", o) for o in re.findall(r"^\s+([vs]_\w+|global_\w+|image_\w+|scratch_\w+)", dis, re.M)]


def test_gfx950_reference_build_exists_and_exports_the_kernels(built):
    assert os.path.isfile(HSACO) and os.path.isfile(HSACO_NC)
    notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", HSACO], capture_output=True, text=True).stdout
    for k in ("trace", "retrace", "createScene", "ref950_sample_frame"):
        assert re.search(r"\.name:\s+%s\b" % k, notes), k
    # the reference's own kernels keep their image arguments; the wrapper writes a buffer
    assert notes.count(".value_kind:     image") >= 4


@pytest.mark.parametrize("path", [HSACO, HSACO_NC])
def test_real_rocm_builtins_are_not_the_ieee_plain_contract(built, path):
    o = ops(path, "ref950_sample_frame")
    assert len(o) > 1000
    # division: reciprocal-based (OpenCL's 2.5 ulp), never the IEEE v_div_scale / v_div_fmas / v_div_fixup sequence
    assert "v_rcp_f32" in o and "v_div_scale_f32" not in o and "v_div_fixup_f32" not in o
    # normalize: rsq-based
    assert "v_rsq_f32" in o
    # dot / cross / mix: fused multiply-adds from the library's own code — also under -ffp-contract=off, which
    # only governs the KERNEL's source expressions
    assert sum(x.startswith(("v_fma", "v_fmac", "v_pk_fma")) for x in o) > 50
    # sqrt: the hardware instruction with range scaling, no correction step; no scratch, no spills
    assert "v_sqrt_f32" in o and not any(x.startswith("scratch_") for x in o)
    # the index hash is evaluated in double in every build (raytracer.cl:114: float x double literal)
    assert "v_mul_f64" in o and "v_cvt_f64_f32" in o and "v_cvt_u32_f64" in o


def test_contraction_flag_changes_only_the_kernels_own_expressions(built):
    a, b = ops(HSACO, "ref950_sample_frame"), ops(HSACO_NC, "ref950_sample_frame")
    fa = sum(x.startswith(("v_fma", "v_fmac", "v_pk_fma")) for x in a)
    fb = sum(x.startswith(("v_fma", "v_fmac", "v_pk_fma")) for x in b)
    assert fb < fa and fb > 50, (fa, fb)
