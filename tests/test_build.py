"""Build-time properties of the kernel library that no numerics test can see reliably."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_tile_draw_register_is_left_alone(tmp_path):
    """The 256x256 GEMM receives its next-tile draw asynchronously in v255 (inline asm).  That is sound only while the
    compiler never allocates v255 in those kernels: every instance that draws must mention v255 exactly twice (the
    atomic and the move that reads it), as a single register and not inside a register range."""
    out = tmp_path / "gemm.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                           os.path.join(ROOT, "multimae_amd", "csrc", "gemm.hip"), "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    text = out.read_text()
    kernels = re.findall(r"^(_Z17gemm_nt_pp_kernel\w+):[^\n]*\n(.*?)s_endpgm", text, flags=re.S | re.M)
    assert len(kernels) >= 14, [k for k, _ in kernels]
    drawing = 0
    for name, body in kernels:
        if "global_atomic_add v255" not in body:
            continue  # the instance that keeps the static stride
        drawing += 1
        lines = [ln.strip() for ln in body.splitlines() if re.search(r"\bv255\b|:255\]", ln)]
        assert len(lines) == 2, (name, lines)
        assert lines[0].startswith("global_atomic_add v255,") and re.fullmatch(r"v_mov_b32 v\d+, v255", lines[1]), (name, lines)
        draw, move = body.index("global_atomic_add v255"), body.index(", v255")
        assert draw < move, name
    assert drawing >= 13
