"""-m gpu: the stand-in policy's rollout shadows (dcmrta_amd/policy.py::rollout_copy) and the fast paths that only exist on the
GPU (SDPA kernels, fp16 / bf16 GEMMs) against the fp32 module, within stated tolerances."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 3e-2), (torch.bfloat16, 2e-1)])
def test_rollout_shadow_tracks_fp32(gpu_device, dtype, tol):
    """Log-probabilities of the low-precision shadow vs the fp32 net on the same inputs: max abs error on unmasked actions below
    `tol` (measured: 8e-3 fp16, 6e-2 bf16 at random init), masked actions stay at the -1e4 floor, probabilities sum to 1,
    and the shadow follows a weight update after sync_rollout_copy."""
    from dcmrta_amd.policy import AttentionNet
    torch.manual_seed(0)
    B, A, T = 512, 20, 50
    net = AttentionNet().to(gpu_device).eval()
    tasks, agents = torch.rand(B, T + 1, 5, device=gpu_device), torch.rand(B, A, 6, device=gpu_device)
    mask = torch.rand(B, T + 1, device=gpu_device) < 0.3
    mask[:, 0] = False
    for nopad in (False, True):
        net.assume_no_padding = nopad
        shadow = net.rollout_copy(dtype)
        with torch.no_grad():
            ref, got = net(tasks, agents, mask), shadow(tasks, agents, mask)
        assert got.dtype == torch.float32 and torch.isfinite(got).all()
        assert (got[~mask] - ref[~mask]).abs().max().item() < tol
        assert (got[mask] < -9000).all()
        assert torch.allclose(got.exp().sum(1), torch.ones(B, device=gpu_device), atol=1e-4)
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.01 * torch.randn_like(p))
        before = shadow(tasks, agents, mask)
        net.sync_rollout_copy(shadow)
        after, ref2 = shadow(tasks, agents, mask), net(tasks, agents, mask)
    assert (after[~mask] - ref2[~mask]).abs().max().item() < tol
    assert (before[~mask] - ref2[~mask]).abs().max().item() > (after[~mask] - ref2[~mask]).abs().max().item()


def test_gpu_forward_matches_reference_fixture(gpu_device, golden_dir):
    """tests/golden/policy_kat.npz (a seeded reference AttentionNet, padded inputs) through the GPU kernels (SDPA, group-norm):
    the same tolerance as the CPU test."""
    import os
    from dcmrta_amd.policy import AttentionNet, load_reference_state_dict
    z = np.load(os.path.join(golden_dir, "policy_kat.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}
    net = load_reference_state_dict(AttentionNet(6, 5, 16).eval(), sd).to(gpu_device)
    with torch.no_grad():
        lp = net(*(torch.from_numpy(z[k]).to(gpu_device) for k in ("in_tasks", "in_agents", "in_mask"))).cpu().numpy()
    ok = ~z["in_mask"]
    np.testing.assert_allclose(lp[ok], z["out_logp"][ok], rtol=1e-4, atol=2e-5)
    assert (lp[z["in_mask"]] < -9000).all()
