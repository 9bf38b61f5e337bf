"""No kernel of libcfx.so may use scratch memory (register spills).

hipcc's -Rpass-analysis=kernel-resource-usage remarks for every csrc/*.hip, compiled with the product flags (tools/resource_usage.py);
cross-compiled, no GPU needed.  A spill on a default-path kernel is extra HBM traffic nobody planned for ("weak" 2:
k_int2_compress_gated<4> 16 B/lane, k_minmax_layer<int4,8> 12, k_minmax_layer<int8,8> 80).
"""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))


@pytest.fixture(scope="module")
def rows():
    import resource_usage
    return resource_usage.collect()


def test_every_kernel_is_seen(rows):
    names = " ".join(k["demangled"] for k in rows)
    for want in ("k_absmean_compress", "k_int2_compress_gated", "k_minmax_layer", "k_topk_layer", "k_lrs", "k_binary_dequant",
                 "k_int2_dequant", "k_int8_dequant", "k_int4_dequant", "k_attn_merge"):
        assert want in names, want
    assert len(rows) > 60


def test_no_kernel_spills_to_scratch(rows):
    bad = [(k["demangled"], k["scratch"]) for k in rows if k.get("scratch", 0) != 0]
    assert not bad, f"kernels with scratch (bytes per lane): {bad}"


def test_layer_kernels_keep_two_workgroups_a_cu(rows):
    # the one-launch layer forms count on 2 workgroups of 512 threads per CU: <= 128 VGPRs, <= 80 KB of LDS
    for k in rows:
        if any(n in k["demangled"] for n in ("k_absmean_compress", "k_int2_compress_gated", "k_minmax_layer")):
            assert k["vgpr"] + k.get("agpr", 0) <= 128 and k["lds"] <= 80 * 1024, k


def test_one_bit_layer_kernel_leaves_room_for_a_collective_kernel(rows):
    # the collective form of the exchange layer needs RCCL's kernel (256 threads x 280 VGPRs) placed BESIDE a waiting reconstruction
    # workgroup: two waves a SIMD of at most 104 registers leave 512 - 208 = 304 (include/cfx.h, cfx_plan_add_exchange_layer;
    # tests/test_gpu_bench.py::test_bench_with_a_collective_kernel_of_rccl_footprint is the run-time check)
    ks = [k for k in rows if k["demangled"].startswith("k_absmean_compress<true, 4, true")]
    assert ks
    for k in ks:
        assert k["vgpr"] <= 104 or k["demangled"].endswith("true>"), k      # (the <.., true> instantiation carries the developer stamps)
