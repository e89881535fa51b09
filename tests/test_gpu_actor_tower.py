"""GPU tests of the fused actor-tower kernels (csrc/pmx_actor.hip) against plain PyTorch float32 of the same tower
(pacman_mappo_resnet.py:49-67, 104-113).

Two references:
  * `emulated`: float32 torch ops with the kernel's roundings made explicit (bf16 weights, bf16 activations between layers,
    the convolution output rounded to bf16 before GroupNorm -- what bf16 autocast computes).  The forward must agree to
    within accumulation order: max error <= 4 bf16 ulps of the largest feature, mean error <= 2e-4 of it.
  * `exact`: the module itself in float32, no rounding anywhere.  Tolerance = what bf16 costs: 3e-2 relative (Frobenius).
Gradients: relative Frobenius error per parameter tensor against autograd through the emulated reference (straight-
through rounding) evaluated ON THE CPU IN FLOAT64, <= 3e-2; the kernel also rounds the gradients between layers to bf16,
as autocast does.  (The CPU is used on purpose: torch 2.10+rocm7.0's native GPU group_norm backward returns GroupNorm
weight/bias gradients that are 100 % off for this shape once the batch exceeds a few hundred samples -- found while
writing this test, tools/actor_debug.py prints both against float64; the product never calls that kernel.)"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.to(torch.bfloat16).float()


def _ste(x):
    return x + (_bf(x) - x).detach()


def _model(H, W, seed=0):
    from pmx import mappo
    torch.manual_seed(seed)
    m = mappo.MAPPOAgent((8, H, W)).cuda()
    with torch.no_grad():                      # non-trivial biases and GroupNorm affine parameters
        for p in m.actor_backbone.parameters():
            if p.dim() == 1:
                p.add_(0.3 * torch.randn_like(p))
    return m


def _obs(B, H, W, seed=1):
    g = torch.Generator(device="cuda").manual_seed(seed)
    o = (torch.rand(B, 8, H, W, device="cuda", generator=g) < 0.25).float()
    o[:, 1] *= torch.randint(1, 6, (B, 1, 1), device="cuda", generator=g).float()     # plane 1 carries 1 + numCarrying
    return o


def emulated_tower(m, obs, ste=False):
    if ste:
        rnd = lambda x: x + (x.float().to(torch.bfloat16).to(x.dtype) - x).detach()
    else:
        rnd = _bf
    bb = m.actor_backbone
    x = obs.to(bb[0].weight.dtype)

    def conv(c, x):
        return rnd(F.conv2d(x, rnd(c.weight), None, padding=1) + c.bias.view(1, -1, 1, 1))
    x = rnd(F.gelu(conv(bb[0], x)))
    x = rnd(F.gelu(conv(bb[2], x)))
    for blk in (bb[4], bb[5], bb[6]):
        y = rnd(F.gelu(F.group_norm(conv(blk.conv1, x), 4, blk.gn1.weight, blk.gn1.bias, 1e-5)))
        x = rnd(F.gelu(F.group_norm(conv(blk.conv2, y), 4, blk.gn2.weight, blk.gn2.bias, 1e-5) + x))
    return x                                    # [B, 32, H, W]


def _board(layout):
    """(H, W) of a named layout, or of the 20 x 20 boards mazeGenerator.py:255-264 produces ("maze")."""
    import pmx
    if layout == "maze":
        from pmx import maze_generator
        lay = pmx.Layout.from_text(maze_generator.generate_maze(7))
    else:
        lay = pmx.get_layout(layout)
    return lay.height, lay.width


@pytest.mark.parametrize("layout,B", [("smallCapture", 777), ("tinyCapture", 777), ("bloxCapture", 333), ("maze", 2111)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.uint8])
def test_tower_forward_matches_torch(layout, B, dtype):
    # B: not a multiple of the wavefronts per block (ragged tail); 2 111 samples of a 20 x 20 board exceed the 2 048 blocks the
    # four-waves-per-sample kernel launches, so some blocks walk two samples
    from pmx import actor_tower
    H, W = _board(layout)
    assert actor_tower.tower_supported(H, W)
    m = _model(H, W)
    obs = _obs(B, H, W)
    with torch.no_grad():
        feat = actor_tower.actor_tower(m.actor_backbone, obs.to(dtype))              # [B, HW, 32]
        got = feat.float().permute(0, 2, 1).reshape(B, 32, H, W)
        emu = emulated_tower(m, obs)
        exact = m.actor_backbone[:-1](obs)
    scale = emu.abs().max().item()
    # a value that sits on a bf16 rounding boundary in one layer may differ by an ulp and move a few more downstream:
    # <= 4 ulps of the largest feature anywhere, and far less on average
    assert (got - emu).abs().max().item() <= scale * 2 ** -6, "forward differs from the rounding-exact emulation"
    assert (got - emu).abs().mean().item() <= scale * 2e-4
    rel = ((got - exact).norm() / exact.norm()).item()
    assert rel < 3e-2, rel
    # the model's logits through the fused path == through the library path, to bf16 accuracy
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        a = m.logits(obs.to(torch.bfloat16)).float()
        m.fused_tower = False
        b = m.logits(obs.to(torch.bfloat16)).float()
        m.fused_tower = True
    assert ((a - b).norm() / b.norm()).item() < 5e-2


@pytest.mark.parametrize("layout,B", [("smallCapture", 515), ("tinyCapture", 96), ("smallCapture", 3), ("bloxCapture", 67), ("maze", 5),
                                      ("bloxCapture", 1)])
def test_tower_backward_matches_autograd(layout, B):
    from pmx import actor_tower
    H, W = _board(layout)
    m = _model(H, W, seed=3)
    obs = _obs(B, H, W, seed=4)
    g = torch.Generator(device="cuda").manual_seed(5)
    dfeat = torch.randn(B, 32, H, W, device="cuda", generator=g) * 0.1
    params = actor_tower._tower_params(m.actor_backbone)
    # fused
    feat = actor_tower.actor_tower(m.actor_backbone, obs.to(torch.bfloat16))
    loss = (feat.float().permute(0, 2, 1).reshape(B, 32, H, W) * dfeat).sum()
    got = torch.autograd.grad(loss, params)
    # reference: autograd through the emulation with straight-through rounding, float64 on the CPU
    mc = _model(H, W, seed=3).cpu().double()
    want = torch.autograd.grad((emulated_tower(mc, obs.cpu(), ste=True) * dfeat.cpu().double()).sum(),
                               actor_tower._tower_params(mc.actor_backbone))
    names = [n for n, _ in m.actor_backbone.named_parameters()]
    worst = 0.0
    for p, a, b in zip(params, got, want):
        assert a.shape == b.shape and a.dtype == p.dtype
        a, b = a.cpu().double(), b
        rel = ((a - b).norm() / (b.norm() + 1e-12)).item()
        worst = max(worst, rel)
        assert rel < 3e-2, (tuple(p.shape), rel)
    assert worst > 0.0
    assert len(names) == len(params)


@pytest.mark.parametrize("layout,B", [("bloxCapture", 1300), ("smallCapture", 8400)])
def test_tower_backward_of_a_large_batch_is_the_sum_over_its_parts(layout, B):
    """Batches beyond one round of blocks (the data kernel of a 20 x 20 board launches at most 1 024 blocks, the weight kernel
    gives every pair of waves a run of samples): the parameter gradients of the whole batch equal the sum of the gradients of
    its two halves.  (Both halves of the smallCapture case stay above the 4 096 samples from which the data-gradient kernel
    gives a sample two waves, and so above the 1 536 below which the forward kernel splits samples too: the split kernels add
    the GroupNorm sums in another order, which moves a few bf16 roundings and the gradients by ~3e-3.)"""
    from pmx import actor_tower
    H, W = _board(layout)
    m = _model(H, W, seed=11)
    obs = _obs(B, H, W, seed=12).to(torch.uint8)
    g = torch.Generator(device="cuda").manual_seed(13)
    dfeat = (torch.randn(B, H * W, 32, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    params = actor_tower._tower_params(m.actor_backbone)

    def grads(lo, hi):
        feat = actor_tower.actor_tower(m.actor_backbone, obs[lo:hi])
        return torch.autograd.grad((feat.float() * dfeat[lo:hi].float()).sum(), params)
    whole, a, b = grads(0, B), grads(0, B // 2), grads(B // 2, B)
    for w, x, y in zip(whole, a, b):
        ref = x.double() + y.double()
        assert ((w.double() - ref).norm() / (ref.norm() + 1e-12)).item() < 2e-4        # float32 sums in another order


def test_tower_backward_is_deterministic_and_zero_batch():
    import pmx
    from pmx import actor_tower
    lay = pmx.get_layout("smallCapture")
    H, W = lay.height, lay.width
    m = _model(H, W, seed=7)
    obs = _obs(64, H, W, seed=8).to(torch.uint8)
    params = actor_tower._tower_params(m.actor_backbone)
    outs = []
    for _ in range(2):
        feat = actor_tower.actor_tower(m.actor_backbone, obs)
        outs.append(torch.autograd.grad(feat.float().square().sum(), params))
    for a, b in zip(*outs):
        assert ((a - b).norm() / (b.norm() + 1e-12)).item() < 1e-5            # (the LDS float adds of the per-channel sums may differ in order)
    empty = actor_tower.tower_forward(obs[:0], actor_tower.pack_params(params))
    assert empty.shape == (0, H * W, 32)
