"""CPU: the stand-in attention policy honours the reference's tensor contract (attention.py:288-297)."""
import os

import numpy as np
import torch


def test_policy_matches_reference_fixture(golden_dir):
    """tests/golden/policy_kat.npz: a seeded width-16 reference AttentionNet, its weights, padded inputs and outputs."""
    from dcmrta_amd.policy import AttentionNet, load_reference_state_dict
    z = np.load(os.path.join(golden_dir, "policy_kat.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}
    net = load_reference_state_dict(AttentionNet(6, 5, 16).eval(), sd)
    with torch.no_grad():
        lp = net(torch.from_numpy(z["in_tasks"]), torch.from_numpy(z["in_agents"]), torch.from_numpy(z["in_mask"]))
    ref = z["out_logp"]
    ok = ~z["in_mask"]
    np.testing.assert_allclose(lp.numpy()[ok], ref[ok], rtol=1e-4, atol=1e-5)      # unmasked log-probs
    assert (lp.numpy()[z["in_mask"]] < -9000).all()                              # masked logits are -1e4 (attention.py:76)
    np.testing.assert_allclose(np.exp(lp.numpy()).sum(1), 1.0, rtol=1e-5)


def test_policy_contract_shapes():
    from dcmrta_amd.policy import AttentionNet
    net = AttentionNet().eval()
    assert sum(p.numel() for p in net.parameters()) == 2528896 - 6 * 4 * 128 * 128  # reference minus its unused dec_self_attn
    B, A, T = 4, 20, 50
    mask = torch.zeros(B, T + 1, dtype=torch.bool)
    mask[:, 0] = True
    with torch.no_grad():
        lp = net(torch.rand(B, T + 1, 5), torch.rand(B, A, 6), mask)
    assert lp.shape == (B, T + 1) and torch.isfinite(lp).all()
    assert (lp.argmax(1) != 0).all()
