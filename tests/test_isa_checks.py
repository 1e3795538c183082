"""Static ISA checks of the hand-synchronised BPTT kernel (tools/isa_check.py): runs hipcc -S on one source file
(~15 s); skipped where hipcc is absent (the GPU box runs the prebuilt .so and the parity tests instead)."""
import os
import shutil
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_dma_bptt_kernel_isa_has_no_queue_drains_and_no_early_register_use():
    import isa_check
    assert isa_check.main() == []


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_dma_weight_gradient_gemms_keep_their_ring_in_flight():
    import isa_check
    assert isa_check.check_dma_gemms() == []


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_ping_pong_gemms_keep_matrix_and_read_segments_apart():
    import isa_check
    assert isa_check.check_gemm_pp() == []
