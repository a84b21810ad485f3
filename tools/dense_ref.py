"""Reference point, not product: what torch.matmul (hipBLASLt) needs for the DENSE fp16 GEMM of the same shape, weights cycled past the
Infinity Cache, launches captured in a hipGraph.   python tools/dense_ref.py [MxKxN ...]"""
import sys, torch
dev = torch.device("cuda:0")
for spec in (sys.argv[1:] or ["512x4096x4096"]):
    M, K, N = (int(v) for v in spec.split("x"))
    ns = max(2, (320 << 20) // (K * N * 2) + 1)
    ws = [torch.randn(K, N, device=dev).half() * 0.02 for _ in range(ns)]
    x = (torch.randn(M, K, device=dev) * 0.5).half()
    y = torch.empty(M, N, device=dev, dtype=torch.float16)
    for w in ws[:3]:
        torch.matmul(x, w, out=y)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g):
            for i in range(100):
                torch.matmul(x, ws[i % ns], out=y)
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 10.0)
    print(f"dense fp16 {spec}: {best:.2f} us per launch in a graph  ({2.0 * M * N * K / best / 1e6:.0f} TFLOP/s, {ns} weight sets)")
