import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pmx import mappo
S, B = 154, int(os.environ.get("ATTN_B", 8192))
qkv = (torch.randn(S, B, 96, device="cuda")).to(torch.bfloat16).requires_grad_(True)
g = torch.randn(S, B, 32, device="cuda").to(torch.bfloat16)
def t(fn, n=30):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print("fwd us", t(lambda: mappo.attention8_forward(qkv.detach(), want_lse=True)))
def fb():
    qkv.grad = None
    mappo.attention8(qkv).backward(g)
print("fwd+bwd us", t(fb))
