"""bf16 configurations on the CPU emulator: the same kernel sources instantiated with ET = bf16 (bf16 storage, one emulated
v_mfma_f32_16x16x16_bf16 per 16-deep product step), checked against the bf16 statement of the stage oracle (tests/bf16_util.py)."""
import numpy as np
import pytest
import torch

from oracle import stblock_stages as st
from tests.bf16_util import Q, assert_bf16_errors, run_block_case_bf16
from tests.emu_util import bind_emulator

CASES = [
    # c_in, channels, Kt, Ks, gct, act, N, B, T, training
    (1, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 21, 2, 7, True),       # first block: thin tmp_conv1, recomputed gate inputs
    (64, (64, 16, 64), 3, 3, "cheb_graph_conv", "glu", 17, 2, 6, True),      # second block: all four time-stepping kernels
    (64, (64, 16, 64), 3, 3, "graph_conv", "gtu", 35, 1, 5, False),
    (32, (64, 16, 64), 3, 2, "cheb_graph_conv", "glu", 40, 1, 7, True),
]


def test_quant_bf16_is_round_to_nearest_even():
    a = np.array([1.0, 1.00390625, 1.001953125, 1.005859375, -3.1415926, 0.0, 1e-20, 65504.0], dtype=np.float64)
    q = Q(a)
    ref = torch.tensor(a, dtype=torch.float32).to(torch.bfloat16).to(torch.float64).numpy()      # torch rounds RNE as well
    assert np.array_equal(q, ref)
    assert np.array_equal(st.from_bf16_bits(st.to_bf16_bits(a)), q)
    assert np.array_equal(Q(q), q)


@pytest.mark.parametrize("c_in,channels,Kt,Ks,gct,act,N,B,T,training", CASES)
def test_block_bf16_matches_bf16_oracle(c_in, channels, Kt, Ks, gct, act, N, B, T, training):
    bind_emulator()
    stored, f32 = run_block_case_bf16("cpu", c_in, channels, Kt, Ks, gct, act, N, B, T, training)
    assert_bf16_errors(stored, f32)
