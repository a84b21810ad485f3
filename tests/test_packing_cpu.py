"""Host-side packing (quick_amd/packing.py, torch ops) against the oracle and the reference fixtures. CPU only."""
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import golden_files, load_golden
from quick_amd import packing
from quick_amd.fused_utils import QUICK_cat

EXACT = golden_files("exact_")
IDS = [os.path.basename(p)[:-4] for p in EXACT]


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.mark.parametrize("path", EXACT, ids=IDS)
def test_quantize_and_pack_cuda_order_match_reference(path):
    g = load_golden(path)
    G = int(g["G"])
    iw = packing.quantize_intweight(_t(g["weight"]), _t(g["scales_nk"]), _t(g["zeros_nk"]), G)
    assert np.array_equal(iw.numpy(), g["intweight_nk"].T.astype(np.int32))
    s, z = _t(g["scales_nk"]).t().contiguous(), _t(g["zeros_nk"]).t().contiguous().to(torch.int32)
    qw, qs, qz = packing.pack_cuda_order(iw, s, z)
    assert qw.dtype == torch.int32 and qs.dtype == torch.float16 and qz.dtype == torch.int32
    assert np.array_equal(qw.numpy(), g["ref_qweight"])
    assert np.array_equal(qs.numpy().view(np.uint16), g["ref_qscales"].view(np.uint16))
    assert np.array_equal(qz.numpy(), g["ref_qzeros"])
    iw2, s2, z2 = packing.unpack_cuda_order(_t(g["ref_qweight"]), _t(g["ref_qscales"]), _t(g["ref_qzeros"]))
    assert np.array_equal(iw2.numpy(), iw.numpy()) and np.array_equal(z2.numpy(), z.numpy())
    assert torch.equal(s2, s)


@pytest.mark.parametrize("path", [p for p in EXACT if "k64" not in p], ids=[i for i in IDS if "k64" not in i])
def test_mi355x_order_matches_oracle_and_bridges_both_ways(path):
    g = load_golden(path)
    iw, s, z = oracle.unpack_cuda_order(g["ref_qweight"], g["ref_qscales"], g["ref_qzeros"])
    want = oracle.pack_mi355x(iw, s, z)
    got = packing.cuda_to_mi355x(_t(g["ref_qweight"]), _t(g["ref_qscales"]), _t(g["ref_qzeros"]))
    for a, b in zip(got, want):
        assert np.array_equal(a.numpy().view(np.uint8), b.view(np.uint8))
    back = packing.mi355x_to_cuda(*got)
    for a, name in zip(back, ("ref_qweight", "ref_qscales", "ref_qzeros")):
        assert np.array_equal(a.numpy().view(np.uint8), g[name].view(np.uint8))


def test_k_not_multiple_of_128_rejected_for_mi355x_order():
    g = load_golden([p for p in EXACT if "k64" in p][0])
    with pytest.raises(ValueError):
        packing.cuda_to_mi355x(_t(g["ref_qweight"]), _t(g["ref_qscales"]), _t(g["ref_qzeros"]))


def test_quick_cat_matches_reference_and_generalises_to_gqa():
    (path,) = golden_files("quick_cat_")
    g = load_golden(path)
    for opt in ("qweight", "qzeros", "scales"):
        got = QUICK_cat(*[_t(g[f"in{j}_{opt}"]) for j in range(3)], options=opt)
        assert np.array_equal(got.numpy().view(np.uint8), g["cat_" + opt].view(np.uint8))
    # unequal widths (the reference raises): cat(pack(a), pack(b)) == pack(concat(a, b))
    rng = np.random.default_rng(5)
    K, G = 256, 64
    parts = []
    for N in (512, 128, 128):
        iw = rng.integers(0, 16, (K, N), dtype=np.uint8)
        z = rng.integers(0, 16, (K // G, N), dtype=np.uint8)
        s = rng.uniform(0.005, 0.025, (K // G, N)).astype(np.float16)
        parts.append((iw, s, z))
    full = oracle.pack_cuda_order(*[np.concatenate([p[i] for p in parts], axis=1) for i in range(3)])
    packs = [oracle.pack_cuda_order(*p) for p in parts]
    for i, opt in enumerate(("qweight", "scales", "qzeros")):
        got = QUICK_cat(*[_t(p[i]) for p in packs], options=opt)
        assert np.array_equal(got.numpy().view(np.uint8), full[i].view(np.uint8))
    with pytest.raises(ValueError):
        QUICK_cat(_t(packs[0][0]), options="qweight")
    with pytest.raises(ValueError):
        QUICK_cat(_t(packs[0][0]), _t(packs[1][0]), options="bogus")
