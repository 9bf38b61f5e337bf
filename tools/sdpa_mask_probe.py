"""Developer probe: PyTorch-ROCm SDPA at the FLUX ring-block shape under CU-masked streams, alone and beside a bandwidth hog.

Answers, for the exchange-lane design: how long is one attention block, how many CUs does it need, and
what does a streaming kernel confined to a disjoint CU set cost it?
Run on the GPU box:  python tools/sdpa_mask_probe.py
"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch

from compactfusion_amd import _lib, codecs as K


def hip_lib():
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                return ctypes.CDLL(line.split()[-1])
    return ctypes.CDLL("libamdhip64.so")


hip = hip_lib()


def masked_stream(bits):
    """bits: iterable of CU-mask bit indices (0..255)"""
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b >> 5] |= 1 << (b & 31)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return torch.cuda.ExternalStream(s.value)


dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
lib = _lib.load()
ctx = K.context(0)
N, H, D = 544, 24, 128
g = torch.Generator(device=dev).manual_seed(0)
q, k, v = (torch.randn(1, H, N, D, device=dev, dtype=torch.float16, generator=g) for _ in range(3))
big_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
big_b = torch.empty(256 << 20, dtype=torch.uint8, device=dev)


def sdpa():
    return torch.ops.aten._scaled_dot_product_flash_attention(q, k, v, 0.0, False, False, scale=D ** -0.5)


def time_on(stream, n=200, hog_stream=None):
    with torch.cuda.stream(stream):
        for _ in range(20):
            sdpa()
    torch.cuda.synchronize()
    if hog_stream is not None:
        for _ in range(40):
            lib.cfx_copy_probe(ctx, big_a.data_ptr(), big_b.data_ptr(), 256 << 20, hog_stream.cuda_stream)   # ~70 us each at 7 TB/s
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        e0.record()
        for _ in range(n):
            sdpa()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def hog_rate(stream, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lib.cfx_copy_probe(ctx, big_a.data_ptr(), big_b.data_ptr(), 256 << 20, stream.cuda_stream)
    torch.cuda.synchronize()
    e0.record(stream)
    for _ in range(n):
        lib.cfx_copy_probe(ctx, big_a.data_ptr(), big_b.data_ptr(), 256 << 20, stream.cuda_stream)
    e1.record(stream)
    torch.cuda.synchronize()
    return 2 * (256 << 20) * n / (e0.elapsed_time(e1) * 1e-3) / 1e12


plain = torch.cuda.Stream(dev)
print(f"SDPA (1,{H},{N},{D}) back to back, plain stream: {time_on(plain):.1f} us per call")
for n_ex in (16, 32):
    ex = masked_stream(range(n_ex))
    comp = masked_stream(range(n_ex, 256))
    t_alone = time_on(comp)
    t_hog_masked = time_on(comp, hog_stream=ex)
    torch.cuda.synchronize()
    t_hog_unmasked_compute = time_on(plain, hog_stream=ex)
    torch.cuda.synchronize()
    print(f"exchange mask = low {n_ex:3d} bits: SDPA on the complement {t_alone:.1f} us; beside a copy hog on the exchange CUs {t_hog_masked:.1f} us; "
          f"SDPA unmasked beside the masked hog {t_hog_unmasked_compute:.1f} us; hog alone on the mask {hog_rate(ex):.2f} TB/s")
    torch.cuda.synchronize()
# whole XCDs for the exchange: its traffic then stays out of the compute XCDs' L2s (bit i = CU i // 8 of XCD i % 8)
for xcds in ((0,), (0, 4), (0, 1)):
    ex_bits = [i for i in range(256) if (i % 8) in xcds]
    co_bits = [i for i in range(256) if (i % 8) not in xcds]
    ex, comp = masked_stream(ex_bits), masked_stream(co_bits)
    t_alone = time_on(comp)
    t_hog = time_on(comp, hog_stream=ex)
    torch.cuda.synchronize()
    print(f"exchange mask = XCDs {xcds} ({len(ex_bits)} CUs): SDPA on the other XCDs {t_alone:.1f} us; beside a copy hog on the exchange XCDs {t_hog:.1f} us; "
          f"hog alone {hog_rate(ex):.2f} TB/s")
    torch.cuda.synchronize()
# half of the CUs of every XCD's lane share, 24 and 16 CUs
for n_ex in (24,):
    ex = masked_stream(range(n_ex)); comp = masked_stream(range(n_ex, 256))
    print(f"exchange mask = low {n_ex} bits: SDPA on the complement {time_on(comp):.1f} us; beside hog {time_on(comp, hog_stream=ex):.1f} us; hog alone {hog_rate(ex):.2f} TB/s")
    torch.cuda.synchronize()
t_both_plain = time_on(plain, hog_stream=torch.cuda.Stream(dev))
print(f"both unmasked: SDPA beside a copy hog {t_both_plain:.1f} us")
