"""Instruction budget of the MFMA loops, from the compiler's gfx950 assembly (hipcc cross-compiles here: no GPU).

Round 5 found the haloed 3x3 kernel ISSUE-bound: its runtime slab state machine cost 65 scalar + 13 vector instructions per slab, 16
non-MFMA instructions per MFMA on the 64 x 128 tile, where two waves per SIMD hide ~10 (profiles/r5_isa_loop_stats.txt,
profiles/r5_halo_unrolled_taps_ab.txt: -0.7 ms per step once the taps were unrolled).  A change that brings such a state machine back --
or makes a hot kernel spill: a scratch reload inside these loops waits with vmcnt(0) and drains the LDS-DMA ring -- passes every parity
test; this one fails."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")


@pytest.fixture(scope="module")
def conv_pl_loops():
    import isa_loop_stats as S
    return list(S.loops(S.assembly(os.path.join(ROOT, "action-detection_amd", "csrc", "conv_pl.hip")), ""))


def test_haloed_kernel_loop_stays_unrolled(conv_pl_loops):
    rows = [r for r in conv_pl_loops if r[0].startswith("conv_pl9_kernel")]
    assert len(rows) == 30, "2 modes x (6 per-image + 9 plain) tile shapes"
    for name, mfma, salu, valu, lds, vmem, total, meta in rows:
        slabs = 18                                   # two channel groups of nine taps per trip
        assert mfma % slabs == 0 and mfma // slabs in (3, 6, 9, 12, 15, 18), (name, mfma)
        assert salu / slabs <= 10 and valu / slabs <= 4, (name, "scalar / vector instructions per slab", salu / slabs, valu / slabs)
        assert (total - mfma) / mfma <= 6.0, (name, "non-MFMA per MFMA", (total - mfma) / mfma)


def test_conv_kernels_do_not_spill(conv_pl_loops):
    assert len(conv_pl_loops) == 54
    for name, *_rest, meta in conv_pl_loops:
        assert meta["ScratchSize"] == "0", (name, meta)
