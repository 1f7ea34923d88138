import torch, time
x = torch.randn(4096, 4096, device="cuda")
y = torch.empty_like(x)
try:
    evs = [torch.cuda.Event(enable_timing=True, external=True) for _ in range(3)]
    print("external kw ok")
except TypeError as e:
    print("no external kw:", e); evs = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
torch.mm(x, x, out=y); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        evs[0].record()
        torch.mm(x, x, out=y)
        evs[1].record()
        torch.mm(x, x, out=y); torch.mm(x, x, out=y)
        evs[2].record()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("elapsed", evs[0].elapsed_time(evs[1]), evs[1].elapsed_time(evs[2]))
except Exception as e:
    print("capture/elapsed failed:", type(e).__name__, e)
