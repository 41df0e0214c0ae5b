"""The ping-pong GEMM's K loop must stay free of what the compiler adds when registers run out: a spill reload
there is a scratch load followed by `s_waitcnt vmcnt(0)`, i.e. a drain of the whole LDS-DMA stream the schedule
keeps in flight (it cost the fused-LayerNorm kernels 2-4 % before the bias-staging addresses were formed at their
use).  Compiles gemm8.hip to gfx950 assembly (no GPU needed) and reads the kernels the engine launches."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"

# <EPI, OUT_F32, RES_TILED, SINGLE, LAYOUT> of the engine's bf16 step: QKV, FFN-up, LayerNorm GEMMs (first layer / later)
ENGINE_KERNELS = ["ILi0ELb0ELb0ELb0ELi4E", "ILi0ELb0ELb0ELb0ELi0E", "ILi1ELb0ELb0ELb0ELi2E", "ILi1ELb0ELb0ELb0ELi0E",
                  "ILi5ELb0ELb1ELb0ELi5E", "ILi5ELb0ELb0ELb0ELi5E"]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_spill_reload_or_compiler_drain_inside_the_k_loop(tmp_path):
    out = tmp_path / "gemm8.s"
    subprocess.run([HIPCC, "-O3", "-std=c++20", "--offload-arch=gfx950", "-Wno-unused-function", "-S",
                    "--cuda-device-only", "gemm8.hip", "-o", str(out)], cwd=os.path.join(ROOT, "clip_amd", "csrc"),
                   check=True, capture_output=True, timeout=600)
    asm = out.read_text()
    for tag in ENGINE_KERNELS:
        names = re.findall(r"^(_ZN[^\n:]*gemm8_kernel" + tag + r"[^\n:]*):", asm, re.M)
        assert names, f"kernel {tag} is not instantiated any more: update ENGINE_KERNELS"
        start = re.search(r"^" + re.escape(names[0]) + r":", asm, re.M).end()
        body = asm[start:asm.index(".amdhsa_kernel", start)].split("\n")
        mfma = [i for i, line in enumerate(body) if "v_mfma" in line]
        assert len(mfma) >= 256
        inside = range(mfma[0], mfma[-1])
        reloads = [body[i].strip() for i in inside if "scratch_load" in body[i]]
        drains = [body[i].strip() for i in inside
                  if "s_waitcnt" in body[i] and "vmcnt" in body[i] and "ASMSTART" not in body[i - 1]]
        assert not reloads, f"{tag}: spill reloads inside the K loop: {reloads[:3]}"
        assert not drains, f"{tag}: compiler-inserted vmcnt waits inside the K loop: {drains[:3]}"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_attention_key_loop_waits_only_where_the_source_says(tmp_path):
    """attention_kernel<false> streams K/V tiles by LDS-DMA issued as inline assembly; the only vmcnt wait between its
    first and last MFMA must be the written-out one at the top of the key loop (the compiler's own would make a tile
    wait for the NEXT tile's loads), and the kernel must keep 128 registers (4 waves per SIMD) without scratch."""
    out = tmp_path / "attention.s"
    subprocess.run([HIPCC, "-O3", "-std=c++20", "--offload-arch=gfx950", "-Wno-unused-function", "-mllvm",
                    "-amdgpu-mfma-vgpr-form", "-S", "--cuda-device-only", "attention.hip", "-o", str(out)],
                   cwd=os.path.join(ROOT, "clip_amd", "csrc"), check=True, capture_output=True, timeout=600)
    asm = out.read_text()
    names = re.findall(r"amdhsa_kernel (_ZN[^\n ]*attention_kernelILb0E[^\n ]*)", asm)
    assert names
    start = re.search(r"^" + re.escape(names[0]) + r":", asm, re.M).end()
    end = asm.index(".amdhsa_kernel", start)
    body = asm[start:end].split("\n")
    mfma = [i for i, line in enumerate(body) if "v_mfma" in line]
    waits = [(i, body[i].strip()) for i in range(mfma[0], mfma[-1]) if "s_waitcnt" in body[i] and "vmcnt" in body[i]]
    assert all("ASMSTART" in body[i - 1] for i, _ in waits), waits
    assert sum("global_load_lds" in line for line in body) == 8          # two tiles' worth: prologue + loop
    meta = asm[end:asm.index(".end_amdhsa_kernel", end)]
    assert int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1)) <= 128
    assert "scratch_" not in "\n".join(body)
