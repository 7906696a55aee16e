"""The built library's kernels do not spill (CPU test: reads what `csrc/build.sh` recorded while compiling).

`hipcc -Rpass-analysis=kernel-resource-usage` prints, per kernel, its register counts, scratch bytes per lane and spill counts;
build.sh keeps that text in `arxiv_rag_amd/_build/<file>.resources.txt`.  A hot kernel that spills stores its registers to
memory once per wave: the first shipped build of the round-2 attention kernel spilled 16 registers that only its fallback path
needed — 0.4 GB of extra writes per launch, invisible to every parity test (found in the WRITE_SIZE counter pass)."""
import re
from pathlib import Path

import pytest

BUILD = Path(__file__).resolve().parents[1] / "arxiv_rag_amd" / "_build"
PAT = re.compile(r"Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)"
                 r".*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+)", re.S)


def _kernels():
    out = {}
    for f in sorted(BUILD.glob("*.resources.txt")):
        for m in PAT.finditer(f.read_text()):
            out[m.group(1)] = dict(vgprs=int(m.group(2)), agprs=int(m.group(3)), scratch=int(m.group(4)), occupancy=int(m.group(5)),
                                   sgpr_spill=int(m.group(6)), vgpr_spill=int(m.group(7)))
    return out


def test_no_kernel_of_the_build_spills_or_uses_scratch():
    ks = _kernels()
    if not ks:
        pytest.skip("no _build/*.resources.txt (library not built by csrc/build.sh in this tree)")
    assert len(ks) > 40                                      # encoder + search + runtime kernels, all template instances
    # (SGPR spills go to lanes of a VGPR, not to memory: the persistent GEMMs have a few and that is fine)
    bad = {k: v for k, v in ks.items() if v["scratch"] or v["vgpr_spill"]}
    assert not bad, bad


def test_hot_kernels_keep_their_occupancy():
    """the occupancy each hot kernel was designed for (waves per SIMD): attention 4 (two 8-wave blocks per CU), the 256x256
    GEMMs 2 (one 8-wave block, 256 registers), search pass A >= 2"""
    ks = _kernels()
    if not ks:
        pytest.skip("no _build/*.resources.txt")
    att = {k: v for k, v in ks.items() if "attention_tr_kernel" in k and k.endswith("ELi8EEvPKtPtPKiPKfifiPy")}
    assert att and all(v["occupancy"] >= 4 for v in att.values()), att
    gemm = {k: v for k, v in ks.items() if "gemm_8phase" in k}
    assert gemm and all(v["occupancy"] >= 2 for v in gemm.values()), gemm
    srch = {k: v for k, v in ks.items() if "search_groupmax_kernel" in k}
    assert srch and all(v["occupancy"] >= 2 for v in srch.values()), srch
