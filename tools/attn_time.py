"""Times the attention kernels alone: ATTN_S (tokens, default 154), ATTN_B (samples, default 8192), batch-major tensors.
A/B switches are read by the library once per process: PMX_ATTN_FWD_V1=1 (the first forward kernel), PMX_ATTN_BWD_TWO_PASS=1."""
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pmx import mappo
S, B = int(os.environ.get("ATTN_S", 154)), int(os.environ.get("ATTN_B", 8192))
qkv = (torch.randn(B, S, 96, device="cuda")).to(torch.bfloat16).requires_grad_(True)
g = torch.randn(B, S, 32, device="cuda").to(torch.bfloat16)
def t(fn, n=30):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
f = t(lambda: mappo.attention8_forward(qkv.detach(), want_lse=True, batch_major=True))
def fb():
    qkv.grad = None
    mappo.attention8(qkv, True).backward(g)
fbt = t(fb)
print(f"S {S} B {B} v1={os.environ.get('PMX_ATTN_FWD_V1')} two_pass={os.environ.get('PMX_ATTN_BWD_TWO_PASS')}: fwd {f:.1f} us, fwd+bwd {fbt:.1f} us, bwd ~{fbt - f:.1f} us")
