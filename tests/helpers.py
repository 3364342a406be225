"""Shared helpers for the test-suite (fixture loading, oracle config from a fixture)."""
import ast
import os

import numpy as np
import torch

from oracle import stgcn_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def cfg_from_fixture(fx) -> orc.OracleConfig:
    return orc.OracleConfig(Kt=int(fx["cfg_Kt"]), Ks=int(fx["cfg_Ks"]), n_his=int(fx["cfg_n_his"]),
                            act_func=str(fx["cfg_act"]), graph_conv_type=str(fx["cfg_gct"]),
                            droprate=float(fx["cfg_droprate"]), blocks=ast.literal_eval(str(fx["cfg_blocks"])))


def real_gso(key):
    return np.load(os.path.join(GOLDEN, "gso_real.npz"))[key]


def synth_gso(n, seed):
    """tests/golden/make_golden.py:synth_gso, restated: operators too large to store are regenerated from (n, seed)."""
    rs = np.random.RandomState(seed)
    a = rs.uniform(-1.0, 1.0, size=(n, n)) * (rs.uniform(size=(n, n)) < 0.5)
    a = a / max(1.0, np.abs(np.linalg.eigvals(a)).max())
    return a.astype(np.float32)


def fixture_gso(name, fx):
    if "gso" in fx:
        return fx["gso"]
    if name == "big600_ks4_f32":
        return synth_gso(600, 7)
    return {"metrla_c2_f32": real_gso("metr_la.cheb_sym_norm_lap"),
            "metrla_c2_b32_f32": real_gso("metr_la.cheb_sym_norm_lap"),          # BASELINE.json configs[1] at its stated batch size
            "pemsbay_c3_b64_f32": real_gso("pems_bay.cheb_sym_norm_lap"),        # configs[2] at its stated batch size
            "pemsd7m_c1_f32": real_gso("pemsd7_m.sym_renorm_adj")}[name]


def fixture_params(fx, cfg, dtype):
    p = orc.random_params(cfg, int(fx["n_vertex"]), seed=int(fx["seed"]), dtype=dtype)
    s, a = orc.param_checksums(p)
    assert abs(s - fx["param_checksum"][0]) <= 1e-6 * max(1.0, abs(a)), "parameter RNG stream drifted: regenerate fixtures"
    assert abs(a - fx["param_checksum"][1]) <= 1e-6 * max(1.0, abs(a))
    return p


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))
