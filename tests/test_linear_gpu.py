"""csrc/linear.hip (include/flexnet.h: flexnet_linear2) — the shared part of the centralised critic's first layer,
mlp_critic.py:25-26 on maddpg.py:33-54's input, as one matrix-core launch in exact fp32 (weights stationary in registers,
round 5) — against an fp64 product.
Tolerance: fp32 accumulation over k <= 740 terms of O(1) products: 3e-7 * sqrt(k) * scale relative to the fp64 result."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(rows, k1, k2, ldw, c2, seed=0, pad1=0, pad2=0):
    from safe_marl_amd.nets import critic_first_layer
    g = torch.Generator(device="cuda").manual_seed(seed)
    W = torch.randn(64, ldw, device="cuda", generator=g) * 0.3
    bias = torch.randn(64, device="cuda", generator=g)
    x1 = torch.randn(rows, k1 + pad1, device="cuda", generator=g)[:, :k1]          # (pad: a row-strided view)
    x2 = torch.randn(rows, max(k2 + pad2, 4), device="cuda", generator=g)[:, :k2]
    with torch.no_grad():
        got = critic_first_layer(bias, x1, x2, W, c2)
    want = bias.double() + x1.double() @ W[:, :k1].double().t() + x2.double() @ W[:, c2:c2 + k2].double().t()
    return got, want


@pytest.mark.parametrize("rows,k1,k2,ldw,c2,pad1,pad2", [
    (32768, 720, 20, 745, 725, 0, 0),        # MADDPG, 5 agents: the update batch
    (65536, 720, 20, 745, 725, 0, 0),        # the largest batch the kernel is used for (nets.LINEAR2_MAX_ROWS)
    (36864, 432, 12, 447, 435, 0, 0),        # 3 agents (BASELINE config 3)
    (12291, 720, 20, 746, 725, 0, 0),        # ragged batch, MATD3's extra flag column in W
    (9000, 144, 4, 149, 145, 8, 4),          # one agent's block, row-strided inputs
])
def test_matches_fp64_product(rows, k1, k2, ldw, c2, pad1, pad2):
    from safe_marl_amd import util
    util.FALLBACKS.pop("critic_fc1", None)
    got, want = _run(rows, k1, k2, ldw, c2, pad1=pad1, pad2=pad2)
    assert "critic_fc1" not in util.FALLBACKS                  # the kernel really ran
    assert got.shape == (rows, 64) and torch.isfinite(got).all()
    scale = want.abs().max().item()
    assert (got.double() - want).abs().max().item() < 3e-7 * np.sqrt(k1 + k2) * scale


def test_more_than_768_input_columns_take_the_library_path():
    """8 agents: 1 184 input columns — more weights than eight wavefronts' registers hold; flexnet_linear2 declines
    (FLEXNET_EUNSUPPORTED), the two library GEMMs run and the fallback is counted."""
    from safe_marl_amd import util
    util.FALLBACKS.pop("critic_fc1", None)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got, want = _run(40960, 1152, 32, 1192, 1160)
    assert util.FALLBACKS.get("critic_fc1", 0) == 1
    assert (got.double() - want).abs().max().item() < 1e-4 * want.abs().max().item()


def test_library_path_below_the_row_threshold_and_switch():
    from safe_marl_amd import nets
    got, want = _run(100, 720, 20, 745, 725)
    assert (got.double() - want).abs().max().item() < 1e-4
    nets.CRITIC_FC1_FUSED = False
    try:
        lib, want = _run(32768, 720, 20, 745, 725)
    finally:
        nets.CRITIC_FC1_FUSED = True
    fused, _ = _run(32768, 720, 20, 745, 725)
    assert (lib - fused).abs().max().item() < 1e-4 * want.abs().max().item()
    # bit-reproducible run to run (one fmaf chain per output element, no atomics)
    again, _ = _run(32768, 720, 20, 745, 725)
    assert torch.equal(fused, again)


def test_refuses_what_it_cannot_address():
    import ctypes as C
    from safe_marl_amd import _lib
    lib = _lib.load()
    x = torch.zeros(64, 724, device="cuda"); w = torch.zeros(64, 745, device="cuda"); b = torch.zeros(64, device="cuda")
    out = torch.zeros(64, 64, device="cuda")
    a = _lib.FlexLinear2Args()
    a.rows, a.k1, a.k2, a.ld1, a.ld2, a.ldw, a.c1, a.c2 = 64, 724, 0, 724, 0, 745, 0, 0
    a.x1, a.w, a.bias, a.out = x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.flexnet_linear2(C.byref(a), s) == _lib.FLEXNET_EUNSUPPORTED      # k1 not a multiple of 8
    a.k1, a.ld1 = 720, 700
    assert lib.flexnet_linear2(C.byref(a), s) < 0 and lib.flexnet_linear2(C.byref(a), s) != _lib.FLEXNET_EUNSUPPORTED    # rows overlap
    a.ld1, a.ldw = 724, 700
    assert lib.flexnet_linear2(C.byref(a), s) < 0 and lib.flexnet_linear2(C.byref(a), s) != _lib.FLEXNET_EUNSUPPORTED    # block past the weight's row
