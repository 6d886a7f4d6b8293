"""Does a hipMemsetAsync RECORDED into a HIP graph still write its value when the graph is replayed?

The round-3 observation (NOTES_r03 section 8, NOTES_r04 section 1): with the squeeze sums of an EfficientNet plan zeroed by the
runtime's memset (TFIMM_MEMSET_NODE=1) instead of the library's fill kernel, every replay of the recorded plan had the same
16-byte pattern ADDED to all sums -- a pattern made of host stack addresses (0x7ffe........), different in every process.
This probe takes the product kernels out: graphs that hold nothing but memset nodes (through the same C entry point, so the
call is exactly the product's), optionally with a torch kernel between them, over a few sizes and alignments.

    python tools/probes/memset_node_probe.py
"""
import ctypes as C
import os
import sys

os.environ["TFIMM_MEMSET_NODE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd")):
    sys.path.insert(0, p)
import torch

from tfimm.engine import ffi


def churn(depth):
    """different host stack contents between replays"""
    junk = [depth * 0x0101010101010101 + i for i in range(64)]
    return sum(junk) if depth == 0 else churn(depth - 1) + junk[3]


def trial(sizes, offset, neighbours, value=0, replays=6):
    bases = [torch.empty(n + 8192, dtype=torch.uint8, device="cuda") for n in sizes]
    bufs = [b[offset:offset + n] for b, n in zip(bases, sizes)]
    y = torch.zeros(1 << 16, device="cuda")
    for b in bufs:
        b.fill_(0x55)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        s = torch.cuda.current_stream().cuda_stream
        for b in bufs:
            rc = ffi.lib.tfimm_hip_memset_async(C.c_void_p(b.data_ptr()), value, b.numel(), C.c_void_p(s))
            assert rc == 0
            if neighbours:
                y.add_(1.0)
    bad = []
    want = value & 0xff
    for r in range(replays):
        for b in bufs:
            b.fill_(0x55)
        torch.cuda.synchronize()
        churn(3 + r)
        g.replay()
        torch.cuda.synchronize()
        for i, b in enumerate(bufs):
            nz = int((b != want).sum().item())
            if nz:
                bad.append((r, i, nz, bytes(b[:32].cpu().numpy().tobytes()).hex()))
    return bad


def main():
    print(torch.__version__, torch.version.hip, torch.cuda.get_device_name(0))
    b4 = [256 * c * 8 for c in (96, 48, 144, 24, 192, 32, 336, 56, 960, 160)]     # the squeeze sums of EfficientNet-B4, batch 256
    cases = [("one node, 192 KiB", [196608]), ("one node, 8 bytes", [8]), ("one node, 100 bytes (odd size)", [100]),
             ("ten nodes, the B4 sizes", b4), ("32 nodes of 192 KiB", [196608] * 32)]
    total = 0
    for name, sizes in cases:
        for offset in (0, 8, 256):
            for nb in (False, True):
                for value in (0, 0x3c):
                    bad = trial(sizes, offset, nb, value)
                    total += len(bad)
                    tag = "ok" if not bad else f"WRONG in {len(bad)} (replay, node) pairs; first: replay {bad[0][0]} node {bad[0][1]}: " \
                                               f"{bad[0][2]} bytes, head {bad[0][3]}"
                    print(f"{name:34s} offset {offset:4d} kernels between {int(nb)} value {value:#04x}: {tag}", flush=True)
    print("memset nodes:", "all replays wrote the recorded value" if total == 0 else f"{total} wrong (replay, node) pairs")


if __name__ == "__main__":
    main()
