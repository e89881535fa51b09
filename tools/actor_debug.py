import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, pmx
from pmx import actor_tower, _lib
import test_gpu_actor_tower as T
lay = pmx.get_layout(sys.argv[1] if len(sys.argv) > 1 else "smallCapture")
H, W = lay.height, lay.width
B = int(sys.argv[2]) if len(sys.argv) > 2 else 3
m = T._model(H, W, seed=3)
obs = T._obs(B, H, W, seed=4)
params = actor_tower._tower_params(m.actor_backbone)
g = torch.Generator(device="cuda").manual_seed(5)
dfeat = torch.randn(B, 32, H, W, device="cuda", generator=g) * 0.1
f_inf = actor_tower.tower_forward(obs.to(torch.bfloat16), actor_tower.pack_params(params))
feat = actor_tower.actor_tower(m.actor_backbone, obs.to(torch.bfloat16))
print("train fwd == infer fwd:", torch.equal(f_inf, feat.detach()), "nan in feat", torch.isnan(feat.float()).any().item())
loss = (feat.float().permute(0, 2, 1).reshape(B, 32, H, W) * dfeat).sum()
got = torch.autograd.grad(loss, params)
ref_out = T.emulated_tower(m, obs, ste=True)
want = torch.autograd.grad((ref_out * dfeat).sum(), params)
for i, (a, b) in enumerate(zip(got, want)):
    print(i, tuple(a.shape), "nan", int(torch.isnan(a).sum()), "rel", float((a - b).norm() / (b.norm() + 1e-12)))
# cross-check the torch reference itself on the CPU (float64)
mc = T._model(H, W, seed=3).cpu().double()
pc = actor_tower._tower_params(mc.actor_backbone)
def emu_cpu(m, obs):
    import torch.nn.functional as F
    rnd = lambda x: x + (x.float().to(torch.bfloat16).double() - x).detach()
    bb = m.actor_backbone
    x = obs.double()
    conv = lambda c, x: rnd(F.conv2d(x, rnd(c.weight), None, padding=1) + c.bias.view(1, -1, 1, 1))
    x = rnd(F.gelu(conv(bb[0], x))); x = rnd(F.gelu(conv(bb[2], x)))
    for blk in (bb[4], bb[5], bb[6]):
        y = rnd(F.gelu(F.group_norm(conv(blk.conv1, x), 4, blk.gn1.weight, blk.gn1.bias, 1e-5)))
        x = rnd(F.gelu(F.group_norm(conv(blk.conv2, y), 4, blk.gn2.weight, blk.gn2.bias, 1e-5) + x))
    return x
wc = torch.autograd.grad((emu_cpu(mc, obs.cpu()) * dfeat.cpu().double()).sum(), pc)
for i in (2, 8, 16, 17, 22, 27):
    a, b, c = got[i].cpu().double(), want[i].cpu().double(), wc[i]
    print(i, "kernel-vs-cpu", float((a - c).norm() / c.norm()), "gputorch-vs-cpu", float((b - c).norm() / c.norm()))
