"""The C-ABI shared library loads and exports every symbol include/cfx.h declares (no compute calls: CPU only)."""
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(REPO, "include", "cfx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cfx_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported_and_bound():
    from compactfusion_amd import _lib
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 15
    bound = {name for name, _, _ in _lib.SYMBOLS}
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/cfx.h but not exported by libcfx.so"
        assert sym in bound, f"{sym} declared in include/cfx.h but not bound in compactfusion_amd/_lib.py"
    assert bound <= set(declared), bound - set(declared)
    assert lib.cfx_abi_version() == 1


def test_packet_bytes_match_oracle_and_reference_arithmetic():
    """cfx_packet_bytes is host-only arithmetic: compare with the oracle's wire-size formulae (main.py:285-293)."""
    from compactfusion_amd import _lib
    from oracle import ref_np as R
    lib = _lib.load()
    names = {1: "binary", 2: "int2", 3: "int4", 4: "int8", 5: "topk"}
    for (N, C) in [(544, 3072), (4096, 1152), (1024, 1152), (4448, 3072), (512, 1536), (64, 256)]:
        for cid, name in names.items():
            for param in ((1, 2, 4, 8, 16) if cid == 5 else (0,)):
                if cid == 5 and (N * C) % 1024:
                    assert lib.cfx_packet_bytes(cid, N, C, param) == 0
                    continue
                assert lib.cfx_packet_bytes(cid, N, C, param) == 2 * R.packet_halves(name, N, C, param), (name, N, C, param)
    assert lib.cfx_packet_bytes(1, 544, 3072, 0) == 216128
    assert lib.cfx_packet_bytes(4, 4096, 1152, 0) == 4723200
    assert lib.cfx_packet_bytes(1, 8, 20, 0) == 0                    # C % 8
    assert lib.cfx_packet_bytes(3, 7, 64, 0) == 0                    # odd N for int4
    assert lib.cfx_packet_bytes(77, 8, 64, 0) == 0
    assert lib.cfx_workspace_bytes(1, 544, 3072, 0, 2) > 0
    assert lib.cfx_workspace_bytes(1, 544, 3072, 0, 17) == 0


def test_error_paths_without_gpu():
    """Argument validation happens before any HIP call."""
    from compactfusion_amd import _lib
    lib = _lib.load()
    ctx = lib.cfx_create(0)
    assert ctx
    items = (_lib.CompItem * 1)()
    assert lib.cfx_compress_batch(ctx, 1, 8, 64, 0, 0, 1, items, None, 0, None) == -1          # null x
    assert lib.cfx_compress_batch(ctx, 1, 8, 20, 0, 0, 1, items, None, 0, None) == -2          # bad shape
    assert lib.cfx_compress_batch(ctx, 42, 8, 64, 0, 0, 1, items, None, 0, None) == -4         # bad codec
    assert lib.cfx_compress_batch(ctx, 1, 8, 64, 0, 0, 0, items, None, 0, None) == -5          # bad batch
    items[0] = _lib.CompItem(0x1002, None, None, 0x2000)
    assert lib.cfx_compress_batch(ctx, 1, 8, 64, 0, 0, 1, items, None, 0, None) == -3          # misaligned x
    items[0] = _lib.CompItem(0x1000, None, None, 0x2000)
    assert lib.cfx_compress_batch(ctx, 1, 8, 64, 0, 0, 1, items, None, 0, None) == -7          # workspace missing
    assert b"workspace" in lib.cfx_last_error_string(ctx)
    lib.cfx_destroy(ctx)


def test_cpu_tensors_are_refused():
    import torch
    from compactfusion_amd import codecs as K
    from compactfusion_amd._lib import CfxError
    with pytest.raises(CfxError):
        K.compress(1, torch.zeros(8, 64, dtype=torch.float16), None, 8, 64)


def test_plan_building_without_gpu():
    """cfx_plan_* argument handling (no launches): op indices, copy, bounds."""
    from compactfusion_amd import _lib
    lib = _lib.load()
    ctx = lib.cfx_create(0)
    plan = lib.cfx_plan_create(ctx)
    assert plan and lib.cfx_plan_size(plan) == 0
    c = (_lib.CompItem * 2)(_lib.CompItem(0x1000, 0x2000, None, 0x3000), _lib.CompItem(0x4000, 0x5000, None, 0x6000))
    d = (_lib.DecompItem * 16)(*[_lib.DecompItem(0x3000, 0x7000 + 0x1000 * i, 0x7000 + 0x1000 * i) for i in range(16)])
    assert lib.cfx_plan_add_compress(plan, 1, 544, 3072, 0, 0, 2, c, 0x9000, 1 << 20) == 0
    assert lib.cfx_plan_add_decompress(plan, 1, 544, 3072, 0, 16, d) == 1
    assert lib.cfx_plan_add_decompress(plan, 1, 544, 3072, 0, 17, d) == -5            # batch too large
    assert lib.cfx_plan_add_compress(plan, 1, 544, 3077, 0, 0, 2, c, 0x9000, 1 << 20) == -2   # bad shape
    assert lib.cfx_plan_size(plan) == 2
    # low-rank layer as plan ops
    import ctypes
    q0 = (ctypes.c_void_p * 2)(0xa000, 0xb000)
    assert lib.cfx_plan_add_lr_compress(plan, 0, 544, 3072, 8, 1, 2, c, q0, 0x9000, 1 << 20) == 2
    assert lib.cfx_plan_add_lr_decompress(plan, 0, 544, 3072, 8, 14, d, 0x9000, 1 << 20) == 3
    assert lib.cfx_plan_add_lr_compress(plan, 0, 544, 3072, 7, 1, 2, c, q0, 0x9000, 1 << 20) == -2      # odd rank
    assert lib.cfx_plan_add_lr_compress(plan, 1, 544, 3072, 12, 1, 2, c, q0, 0x9000, 1 << 20) == -2     # LOW_RANK_Q: rank % 8
    assert lib.cfx_plan_add_lr_decompress(plan, 0, 544, 3072, 8, 17, d, 0x9000, 1 << 20) == -5           # batch too large
    assert lib.cfx_plan_add_lr_compress(plan, 0, 544, 3072, 8, 1, 2, c, None, 0x9000, 1 << 20) == -1     # no start matrices
    assert lib.cfx_plan_set_input(plan, 2, 1, 0xc000) == 0 and lib.cfx_plan_set_input(plan, 3, 0, 0xc000) == -5
    assert lib.cfx_plan_size(plan) == 4
    other = lib.cfx_plan_create(ctx)
    assert lib.cfx_plan_copy_op(other, plan, 1) == 0 and lib.cfx_plan_copy_op(other, plan, 0) == 1
    assert lib.cfx_plan_copy_op(other, plan, 5) == -5
    assert lib.cfx_plan_add_wait(other, 0) == -5                                       # op 0 is not an all-gather
    assert lib.cfx_plan_set_exchange_stream(other, 7) == -5
    assert lib.cfx_plan_set_exchange_stream(other, 0) == 0
    assert lib.cfx_plan_run(plan, 1, 5, None) == -5                                     # range out of bounds
    # exchange layer (compress ; all-gather ; reconstruct as one op): argument checks come before anything touches a device
    assert lib.cfx_plan_add_exchange_layer(plan, 9, 544, 3072, 0, 1, 2, c, 14, d, None, None, None, 0, 0x9000, 1 << 20) == -4     # unknown codec
    assert lib.cfx_plan_add_exchange_layer(plan, 1, 544, 3072, 0, 1, 2, c, 0, d, None, None, None, 0, 0x9000, 1 << 20) == -5      # nothing to reconstruct
    assert lib.cfx_plan_add_exchange_layer(plan, 1, 544, 3072, 0, 1, 2, c, 17, d, None, None, None, 0, 0x9000, 1 << 20) == -5     # batch too large
    assert lib.cfx_plan_add_exchange_layer(plan, 1, 544, 3072, 0, 1, 2, c, 14, d, 0x1234, None, None, 0, 0x9000, 1 << 20) == -1   # communicator without buffers
    xl = lib.cfx_plan_create(ctx)
    assert lib.cfx_plan_use_exchange_stream(xl, 0x5678) == 0                            # (a caller's stream: nothing is created here)
    assert lib.cfx_plan_add_exchange_layer(xl, 1, 544, 3072, 0, 1, 2, c, 14, d, None, None, None, 0, 0x9000, 1 << 20) == 0
    assert lib.cfx_plan_add_exchange_layer(xl, 1, 544, 3077, 0, 1, 2, c, 14, d, None, None, None, 0, 0x9000, 1 << 20) == -2      # bad shape
    assert lib.cfx_plan_size(xl) == 1
    assert lib.cfx_plan_add_exchange_layer(xl, 3, 544, 3072, 0, 1, 2, c, 14, d, None, None, None, 0, 0x9000, 1 << 20) == 1       # any codec (in-order form)
    assert lib.cfx_plan_add_exchange_layer(xl, 3, 543, 3072, 0, 1, 2, c, 14, d, None, None, None, 0, 0x9000, 1 << 20) == -2      # int4: N even
    assert lib.cfx_plan_set_input(xl, 0, 1, 0xc000) == 0 and lib.cfx_plan_set_input(xl, 0, 2, 0xc000) == -5     # activations re-pointed per call
    # ... and its collective-free form: flags are checked before anything is allocated
    pf = (ctypes.c_void_p * 2)(0xd000, 0xd040)
    assert lib.cfx_plan_add_exchange_layer_p2p(xl, 1, 544, 3072, 0, 1, 2, c, 14, d, None, 2, pf, 0x9000, 1 << 20) == -5          # no own flag
    assert lib.cfx_plan_add_exchange_layer_p2p(xl, 1, 544, 3072, 0, 1, 2, c, 14, d, 0xe000, 16, pf, 0x9000, 1 << 20) == -5       # too many peers
    bad = (ctypes.c_void_p * 2)(0xd000, 0xd042)
    assert lib.cfx_plan_add_exchange_layer_p2p(xl, 1, 544, 3072, 0, 1, 2, c, 14, d, 0xe000, 2, bad, 0x9000, 1 << 20) == -1       # misaligned peer flag
    assert lib.cfx_ipc_open(ctx, None, None) == -1 and lib.cfx_ipc_close(ctx, None) == -1 and lib.cfx_ipc_free(ctx, None) == -1
    lib.cfx_plan_destroy(xl)
    lib.cfx_plan_destroy(other)
    lib.cfx_plan_destroy(plan)
    lib.cfx_destroy(ctx)


def test_context_setters_replace_the_environment_switches():
    """every behaviour switch of the library is a setter on the context (include/cfx.h); the one environment variable it reads is HIP's"""
    import glob
    from compactfusion_amd import _lib
    lib = _lib.load()
    ctx = lib.cfx_create(0)
    assert lib.cfx_set_stats_rows(ctx, 64) == 0 and lib.cfx_set_stats_rows(ctx, -1) == -5 and lib.cfx_set_stats_rows(ctx, 0) == 0
    assert lib.cfx_set_gated_launch(ctx, 0) == 0 and lib.cfx_set_gated_launch(ctx, 1) == 0
    assert lib.cfx_set_lr_chain(ctx, 1) == 0 and lib.cfx_set_lr_chain(ctx, 3) == -5 and lib.cfx_set_lr_chain(ctx, 0) == 0
    assert lib.cfx_set_lr_decode(ctx, 2) == 0 and lib.cfx_set_lr_decode(ctx, 3) == -5 and lib.cfx_set_lr_decode(ctx, 0) == 0
    assert lib.cfx_set_allow_shared_queues(ctx, 1) == 0 and lib.cfx_set_allow_shared_queues(ctx, 0) == 0
    assert lib.cfx_set_fused_finalize(ctx, 0) == 0 and lib.cfx_set_fused_finalize(ctx, 1) == 0
    assert lib.cfx_ipc_memory_kind(ctx) == 0 and lib.cfx_ipc_memory_kind(None) == -1
    assert lib.cfx_hw_queues_ok() in (0, 1)
    plan = lib.cfx_plan_create(ctx)
    assert lib.cfx_plan_set_pipe_unit_layers(plan, 3) == 0 and lib.cfx_plan_set_pipe_unit_layers(plan, 8) == -5
    lib.cfx_plan_destroy(plan)
    lib.cfx_destroy(ctx)
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "compactfusion_amd", "csrc")
    n = sum(open(f).read().count("getenv(") for f in glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.h")))
    assert n == 1, "libcfx reads exactly one environment variable (GPU_MAX_HW_QUEUES, the HIP runtime's)"


def test_lowrank_sizes_without_gpu():
    from compactfusion_amd import _lib
    lib = _lib.load()
    assert lib.cfx_lr_packet_bytes(0, 544, 3072, 8) == (544 + 3072) * 8 * 2            # 57 856 B
    assert lib.cfx_lr_packet_bytes(1, 544, 3072, 32) == 2 * (544 * 32 // 4 + 64 + 3072 * 32 // 4 + 64)
    assert lib.cfx_lr_packet_bytes(0, 544, 3072, 7) == 0 and lib.cfx_lr_packet_bytes(1, 544, 3072, 12) == 0
    assert lib.cfx_lr_packet_bytes(0, 544, 3072, 34) == 0
    assert lib.cfx_lr_workspace_bytes(0, 544, 3072, 8, 2) > 2 * 544 * 3072 * 2


def test_header_is_plain_c():
    """include/cfx.h is the drop-in boundary: it must compile as C99 on its own (no C++ constructs, no torch / HIP types)"""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if not gcc:
        import pytest
        pytest.skip("no gcc")
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "cfx.h")
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", hdr], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_product_library_has_no_developer_entry_points():
    """include/cfx_dev.h (per-workgroup phase stamps, the launch-tag test hook, early exits of the compress kernel) belongs to libcfx_dev.so
    - the same sources with -DCFX_DEV_PROBES.  The product library exports none of it, binds none of it, and its kernels take no probe
    argument (csrc/cfx_internal.h: `Probe` is an empty type without the macro); the developer library exports exactly what the header
    declares."""
    import re
    import subprocess
    from compactfusion_amd import _lib
    from compactfusion_amd.build import LIB, build_lib
    lib = _lib.load()
    hdr = open(os.path.join(REPO, "include", "cfx_dev.h")).read()
    dev_syms = re.findall(r"^int\s+(cfx_dev_\w+)\s*\(", hdr, flags=re.M)
    assert sorted(dev_syms) == sorted(n for n, _, _ in _lib.DEV_SYMBOLS) and len(dev_syms) == 3
    exported = subprocess.run(["nm", "-D", "--defined-only", LIB], capture_output=True, text=True, check=True).stdout
    assert not re.search(r"debug|cfx_dev_|stamp", exported), "the product library exports a developer symbol"
    for sym in dev_syms:
        assert not hasattr(lib, sym) or os.environ.get("CFX_LIBCFX_PATH"), sym
    assert not any(n.startswith(("cfx_dev", "cfx_debug")) for n, _, _ in _lib.SYMBOLS)
    dev = build_lib(dev_probes=True)
    exported_dev = subprocess.run(["nm", "-D", "--defined-only", dev], capture_output=True, text=True, check=True).stdout
    for sym in dev_syms:
        assert re.search(r"\b%s\b" % sym, exported_dev), f"{sym} declared in include/cfx_dev.h but not exported by libcfx_dev.so"
    # no kernel of the product build mentions a probe: the only `Probe` with a pointer in it sits behind the macro
    src = open(os.path.join(REPO, "compactfusion_amd", "csrc", "cfx_internal.h")).read()
    assert "#ifdef CFX_DEV_PROBES\nstruct Probe {\n    unsigned long long* p;" in src
