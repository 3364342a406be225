"""Launch-time kernel variants that the residency heuristic may pick on a GPU (tile rows of the gated conv, wave count of
the transposed conv) are forced one by one and run through the same emulator stage tests: on the CPU emulator every
variant "fits", so without forcing only the smallest tile would ever be exercised here."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FWD = "tests/test_emu_forward.py"
BWD = "tests/test_emu_backward.py"


def run_subset(env_extra, files, k):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-m", "pytest", *files, "-x", "-q", "-p", "no:cacheprovider", "-k", k],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("tr", [32, 48, 64])
def test_gated_conv_tile_rows(tr):
    # 1-channel first layer (scalar staging), 64-channel layers with / without an Align epilogue, ragged last tile
    run_subset({"STGCN_TCONV_TR": str(tr)}, [FWD], "21-2-7 or 17-1-6 or 35-1-5")


@pytest.mark.parametrize("waves", [4, 8])
def test_transposed_conv_wave_variants(waves):
    run_subset({"STGCN_BWD_DATA_WAVES": str(waves)}, [BWD], "17-2-6")


@pytest.mark.parametrize("parts", [pytest.param("1,1", marks=pytest.mark.full), pytest.param("2,3", marks=pytest.mark.full), "4,2"])
def test_graph_conv_slab_parts(parts):
    # workgroups per (b, t) slab of the graph conv, forward / backward (Chebyshev Ks = 3 and 5, Kipf, 300-node graph)
    run_subset({"STGCN_GC_PARTS": parts}, [FWD], "17-1-6 or 35-1-5 or 9-2-5" + (" or 300-6-12" if parts != "1,1" else ""))
    run_subset({"STGCN_GC_PARTS": parts}, [BWD], "17-2-6 or 35-1-5 or 9-2-5")


@pytest.mark.full
def test_time_complete_conv_tiles():
    # opt-in gated conv with one workgroup per (window, 16 nodes) over all time steps (tconv_fwd3_kernel)
    run_subset({"STGCN_TCONV_V": "3"}, [FWD], "17-1-6 or 35-1-5 or 300-6-12")


@pytest.mark.full
@pytest.mark.parametrize("per_cu", [1, 3])
def test_operator_stationary_graph_conv(per_cu):
    # opt-in graph conv with a wave's operator fragments in registers over several slabs (gconv_fwd_reg_kernel)
    run_subset({"STGCN_GC_REG": str(per_cu)}, [FWD], "17-1-6 or 35-1-5 or 9-2-5")


@pytest.mark.full
def test_head_tap_masked_kernels():
    # the row-tile kernels the output head used before the dense 32 x 256 tiles (tconv_fwd4_kernel) stay selectable
    run_subset({"STGCN_TCONV4": "0"}, ["tests/test_emu_head.py"], "head")


@pytest.mark.parametrize("ntw", [4, 5])
def test_tiled_gemm_column_extents(ntw):
    # workgroup tiles of 128 / 160 GEMM columns of the tiled graph conv's fp32 operator GEMM (the emulator's residency
    # heuristic always picks the 96-column tile): 10 slabs = ragged column tiles for both, Chebyshev and Kipf
    run_subset({"STGCN_GEMM_NTW": str(ntw)}, ["tests/test_emu_gctile.py"], "stage_oracle and (21-2-7 or 35-1-5)")


def test_bf16_gemm_64_deep_steps():
    # the bf16 / bf16x3 operator GEMM with 64-deep pipeline steps (128-B LDS rows, 8-position swizzle); the default is 32
    run_subset({"STGCN_GEMM_BF16_BK": "64"}, ["tests/test_emu_gctile.py"], "rounded or (bf16x3 and graph_conv)")


def test_big_bf16_gemm_32_deep_steps():
    # the 256-row-tile bf16 operator GEMM on its 4-buffer ring of 32-deep steps (the default is 64-deep steps on two buffers)
    run_subset({"STGCN_GEMM_BIG_BK": "32"}, ["tests/test_emu_gctile.py"], "bf16_gemm or tiled_block")


def test_layernorm_slab_statistics_prepass():
    # the stage-per-launch LayerNorm (graphs beyond the fused kernel's 448 nodes; here forced by STGCN_FUSE without bit 2) with its slab
    # statistics from the one-workgroup-per-slab pre-pass (ln_slab_stats_kernel) that big slabs take, forced for every slab
    run_subset({"STGCN_FUSE": str(0x7fffffff & ~2), "STGCN_LN_STATS_MIN_CHUNKS": "1"}, [FWD], "17-1-6 or 35-1-5")
    run_subset({"STGCN_FUSE": str(0x7fffffff & ~2), "STGCN_LN_STATS_MIN_CHUNKS": "1"}, [BWD], "17-2-6")
    # (bf16: the tiled configurations take this LayerNorm; tests/test_emu_bf16.py::test_tiled_block_bf16_matches_bf16_oracle with the pre-pass forced)
    run_subset({"STGCN_LN_STATS_MIN_CHUNKS": "1"}, ["tests/test_emu_bf16.py"], "tiled_block_bf16 and 37-2-6")


def test_reduction_big_table_forms():
    """reduce_kernel's forms for tables of >= 65 536 elements (16-byte gradient / AdamW state accesses of flat jobs, 4 slices for <= 32
    partials -- C5's LayerNorm parameters) forced on the small models of the optimizer / model tests: fused reduce + AdamW against
    torch.optim.AdamW, gradients against the reference's goldens."""
    run_subset({"STGCN_REDUCE_BIG": "1"}, ["tests/test_emu_optim.py"], "trajectory or fused_step_tail or fused_into_the_head")
    run_subset({"STGCN_REDUCE_BIG": "1"}, ["tests/test_emu_model.py"], "golden")


@pytest.mark.parametrize("mask", ["y", "philox"])
def test_dropout_mask_source_of_the_layernorm_backward(mask):
    """Where the backward takes a block's dropout mask from: read off the block output (the default for fp32 and bf16 since round 5: the
    forward stores a dropped element as -0.0 and a kept zero as +0.0, so "dropped iff y is -0.0" is exact) or regenerated with Philox
    (STGCN_HOOK_MASK=philox, for callers that cannot hand the backward the bit-exact forward output) -- fp32 and bf16 blocks in training mode,
    and the whole model (hook epilogues of the next block / the head)."""
    run_subset({"STGCN_HOOK_MASK": mask}, [BWD], "17-2-6-True")
    run_subset({"STGCN_HOOK_MASK": mask}, ["tests/test_emu_bf16.py"], "block_bf16 and 17-2-6")
    run_subset({"STGCN_HOOK_MASK": mask}, ["tests/test_emu_model.py"], "golden and tiny_cheb_f32")


def test_thin_first_layer_row_tile_kernels():
    """STGCN_THIN=0: the thin first layer (K = Kt * c_in <= 4) on the row-tile kernels of rounds 1 - 4 (tconv_fwd_kernel with K padded to 16,
    thin_tc1_bwd_kernel) instead of the wave-per-tile kernels of round 5 (stgcn_kernels_thin.hip.h, the default): fp32, bf16, bf16x3 backward."""
    run_subset({"STGCN_THIN": "0"}, [FWD], "1-channels0 or 2-channels6")
    run_subset({"STGCN_THIN": "0"}, [BWD], "1-channels0 or 2-channels7")
    run_subset({"STGCN_THIN": "0"}, ["tests/test_emu_bf16.py"], "block_bf16 and 21-2-7")
