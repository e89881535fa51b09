import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pmx import mappo
torch.manual_seed(0)
H, W, B = 11, 14, 512
for rep in range(4):
    model = mappo.MAPPOAgent((8, H, W)).cuda()
    L = mappo.PPOLearner(model, autocast_dtype=torch.bfloat16)
    obs = (torch.rand(B, 8, H, W, device="cuda") < 0.2).to(torch.bfloat16)
    merged = (torch.rand(B, 8, H, W, device="cuda") < 0.2).to(torch.bfloat16)
    act = torch.randint(0, 5, (B,), device="cuda"); logp = torch.full((B,), -1.6, device="cuda")
    adv = torch.randn(B, device="cuda"); ret = torch.randn(B, device="cuda")
    L.capture(B, (8, H, W), torch.bfloat16)
    g = []
    for k in range(12):
        st = L.update_minibatch_graph(obs, merged, act, logp, adv, ret)
        torch.cuda.synchronize()
        g.append("%.3g" % float(st["grad_norm"]))
    print(rep, g, bool(torch.isfinite(L.bucket.data).all()), flush=True)
