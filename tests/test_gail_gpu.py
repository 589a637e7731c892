"""GPU: the GAIL-constraint baseline (icrl/gail_utils.py, icrl/gail.py) on the HIP path against the draws and
results recorded from the reference's own GailDiscriminator + GailCallback + PPO (tests/golden/g12_gail.npz).
Tolerances: discriminator metrics / relabelled rewards 2e-5 (fp32 MLP on the GPU vs torch-CPU), advantages 1e-4 (GAE of those
rewards), parameters after 2 x <= 3 epochs x 8 minibatches as in tests/test_ppo_train_gpu.py."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _sub(g, prefix):
    keys = g.files if hasattr(g, "files") else list(g)
    return {k[len(prefix):]: g[k] for k in keys if k.startswith(prefix)}


class _G12Streams:
    """replays g12's draws: per rollout one discriminator permutation, then the PPO epochs' permutations."""

    def __init__(self, g):
        self.noise, self.perms, self.is_disc = g["noise"], g["perms"], g["perm_is_disc"]
        self.i_roll, self.cursor = 0, 0

    def rollout_noise(self, T, N, A):
        n = self.noise[self.i_roll]; self.i_roll += 1
        return n

    def disc_perm(self, itr):
        assert self.is_disc[self.cursor]
        p = self.perms[self.cursor]; self.cursor += 1
        return p[None]

    def permutation(self, epoch, n):
        k = self.cursor + epoch
        return self.perms[k] if k < len(self.perms) and not self.is_disc[k] else np.arange(n)

    def consumed(self, executed):
        self.cursor += int(executed)


def test_gail_two_rollouts_vs_reference(golden):
    from icrl_amd import logger, utils
    from icrl_amd.callbacks import CallbackList
    from icrl_amd.gail import PPO
    from icrl_amd.gail_utils import GailCallback, GailDiscriminator
    g = golden("g12_gail")
    _, T, N, _ = g["noise"].shape
    env = utils.make_train_env("HCWithPos-v0", None, False, 0, N, normalize_cost=False, reward_gamma=0.99)
    assert env.venv is env.unwrapped                      # no cost wrapper in the chain
    streams = _G12Streams(g)
    disc = GailDiscriminator(18, 6, [20], 48, lambda x: 0.01, g["exp_obs"], g["exp_acs"], False, clip_obs=20, eps=1e-5)
    disc.load_state_dict(_sub(g, "d0/"))
    model = PPO("MlpPolicy", env, n_steps=T, batch_size=16, n_epochs=3, target_kl=0.02, seed=0, streams=streams)
    sd = model.policy.state_dict(); sd.update(_sub(g, "w0/")); model.policy.load_state_dict(sd)
    cost_net0 = {k: v.clone() for k, v in model.policy.state_dict().items() if "cost_value" in k}
    cb = GailCallback(disc, False, lambda o, a: (o[..., 0] <= -0.05))
    cb.perms = streams.disc_perm
    seen = []

    class Spy(type(cb)):
        pass
    orig = cb._on_rollout_end

    def spy():
        orig()
        rb = model.rollout_buffer
        seen.append((rb.rewards.cpu().numpy().copy(), rb.reward_advantages.cpu().numpy().copy(), rb.reward_returns.cpu().numpy().copy()))
    cb._on_rollout_end = spy
    model.learn(2 * N * T, callback=CallbackList([cb]))
    assert streams.cursor == len(g["perms"]) and len(seen) == 2
    for it in range(2):
        for k in ("disc_loss", "expert_loss", "nominal_loss", "mean_nominal_preds", "mean_expert_preds"):
            got, ref = cb.history[it]["discriminator/" + k], float(g[f"disc_metrics/{it}/{k}"])
            assert abs(got - ref) < 2e-5 + 2e-4 * abs(ref), (it, k, got, ref)
        assert cb.history[it]["eval/mean_cost"] == float(g["mean_cost"][it])
        assert np.allclose(seen[it][0], g["rewards"][it], rtol=2e-5, atol=2e-5), np.abs(seen[it][0] - g["rewards"][it]).max()
        assert np.allclose(seen[it][1], g["advantages"][it], rtol=1e-4, atol=1e-4)
        assert np.allclose(seen[it][2], g["returns"][it], rtol=1e-4, atol=1e-4)
    lg = logger.Logger.CURRENT.name_to_value
    assert abs(lg["train/approx_kl"] - float(g["log/approx_kl"])) < 2e-5 and abs(lg["train/reward_value_loss"] - float(g["log/value_loss"])) < 2e-4 * max(1, float(g["log/value_loss"]))
    for k, v in _sub(g, "w1/").items():
        got = model.policy.state_dict()[k].numpy()
        assert np.allclose(got, v, rtol=1e-3, atol=2e-5), (k, np.abs(got - v).max())
    for k, v in _sub(g, "d1/").items():
        got = disc.state_dict()[k].numpy()
        assert np.allclose(got, v, rtol=2e-3, atol=2e-5), (k, np.abs(got - v).max())
    # plain PPO never touches the idle cost critic (zero gradients, zero Adam updates) and nu stays ~ 1e-8
    for k, v in cost_net0.items():
        assert torch.equal(model.policy.state_dict()[k], v), k
    assert model.dual.nu().item() < 2e-8
    # reward_function on fresh points, both readings
    disc.load_state_dict(_sub(g, "d1/"))
    assert np.allclose(disc.reward_function(g["probe_obs"], g["probe_acs"]).cpu().numpy(), g["probe_reward"], rtol=2e-5, atol=2e-5)
    assert np.allclose(disc.reward_function(g["probe_obs"], g["probe_acs"], apply_log=False).cpu().numpy(), g["probe_d"], rtol=2e-5, atol=2e-6)


def test_gail_entry_point_short_run(tmp_path):
    """README.md:43 flags of the reference (HC GAIL-constraint: -dl 30 -dlr 0.003 -lc) at a reduced size."""
    from icrl_amd.gail import build_parser, gail
    from icrl_amd.gail_utils import GailDiscriminator
    expert = os.path.join(HERE, "golden/expert_hc.npz")
    argv = ["gail", "-er", "10", "-ep", expert, "-tk", "0.01", "-t", "4096", "-tei", "HCWithPos-v0", "-eei", "HCWithPosTest-v0", "-dl", "30",
            "-dlr", "0.003", "-lc", "-nt", "8", "--n_steps", "128", "-ne", "3", "-s", "1", "-v", "0", "--save_dir", str(tmp_path),
            "--eval_every", "256"]
    cfg = vars(build_parser().parse_args(argv)); cfg.update(rank=0, world_size=1)
    model, disc, hist = gail(types.SimpleNamespace(**cfg), log=None)
    assert model.num_timesteps == 4096 and len(hist) == 4
    assert all(np.isfinite(h["discriminator/disc_loss"]) and 0.0 <= h["discriminator/mean_expert_preds"] <= 1.0 for h in hist)
    assert hist[-1]["discriminator/disc_loss"] < hist[0]["discriminator/disc_loss"]          # the discriminator learns to separate
    assert os.path.exists(tmp_path / "gail_discriminator.pt") and os.path.exists(tmp_path / "best_model.zip")
    again = GailDiscriminator.load(str(tmp_path / "gail_discriminator.pt"))
    for k, v in disc.state_dict().items():
        assert torch.equal(again.state_dict()[k], v), k


def test_cpg_load_gail_cost_is_the_reference_discriminator(golden, tmp_path):
    """`cpg --load_gail -cp <gail_discriminator.pt>` (ref: icrl/cpg.py:54-83): the env cost is D(prev raw obs, clipped action) of
    the loaded discriminator, apply_log=False.  Fixture: g12's discriminator as the REFERENCE left it after two rollouts (d1/*),
    written in the reference's save layout; the costs of the last rollout are recomputed with the oracle's torch-CPU copy."""
    from icrl_amd.cpg import build_parser, cpg
    from icrl_amd.gail_utils import GailDiscriminator
    from oracle import gail as o_gail
    g = golden("g12_gail")
    disc = GailDiscriminator(18, 6, [20], None, lambda x: 0.0, None, None, False, clip_obs=20, eps=1e-5)
    disc.load_state_dict(_sub(g, "d1/"))
    path = str(tmp_path / "gail_discriminator.pt")
    disc.save(path)
    N, T = 4, 64
    argv = ["cpg", "--cn_path", path, "--load_gail", "-tei", "HCWithPos-v0", "-eei", "HCWithPosTest-v0", "-t", str(2 * N * T), "-nt", str(N),
            "--n_steps", str(T), "-ne", "2", "-s", "2", "-v", "0", "--eval_every_rollouts", "2"]
    cfg = vars(build_parser().parse_args(argv)); cfg.update(rank=0, world_size=1, save_dir=None)
    model, hist = cpg(types.SimpleNamespace(**cfg), log=None)
    assert model.num_timesteps == 2 * N * T
    rb = model.rollout_buffer
    od = o_gail.make_disc(18, 6, [20])
    od.load_state_dict(_sub(g, "d1/"))
    prev_obs = rb.orig_observations.cpu().numpy().astype(np.float64)            # raw observation BEFORE each step (float32-rounded in the buffer)
    clipped = np.clip(rb.actions.cpu().numpy(), -1.0, 1.0)
    ref = o_gail.disc_reward(od, prev_obs, clipped, apply_log=False)
    got = rb.orig_costs.cpu().numpy()
    assert got.shape == ref.shape == (T, N)
    assert np.allclose(got, ref, rtol=2e-5, atol=2e-6), np.abs(got - ref).max()
    assert 0.0 < got.min() and got.max() < 1.0 and got.std() > 0
