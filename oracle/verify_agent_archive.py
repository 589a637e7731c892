"""BUILD CONTAINER ONLY (test infrastructure, like oracle/gen_golden.py): load an agent archive written by this build's
`PPOLagrangian.save` with the REFERENCE's own `PPOLagrangian.load` (stable_baselines3/common/base_class.py:564-645,
save_util.py:355-418) and compare its deterministic `predict` with what the HIP path computed on the same 64 observations.

    PYTHONPATH=oracle/ref_shim:/root/reference:/root/reference/custom_envs python -W ignore -m oracle.verify_agent_archive \
        tests/golden/hip_agent_archive.zip tests/golden/hip_agent_archive_expected.npz

The reference is imported unmodified under the third-party stand-ins of oracle/ref_shim (gym is not installed; the archive names
gym.spaces.box.Box by module and name, which is all a pickle needs).  Exit code 0 and a line starting with "VERIFIED" on success."""
import sys

import numpy as np


def main(archive, expected):
    import torch as th
    from stable_baselines3 import PPOLagrangian                      # the reference's class
    from stable_baselines3.common.policies import ActorTwoCriticsPolicy
    import gym
    model = PPOLagrangian.load(archive, device="cpu")                # env=None: prediction only
    assert isinstance(model.policy, ActorTwoCriticsPolicy), type(model.policy)
    assert isinstance(model.observation_space, gym.spaces.Box) and model.observation_space.shape == (18,)
    assert isinstance(model.action_space, gym.spaces.Box) and model.action_space.shape == (6,)
    exp = np.load(expected)
    act, _ = model.predict(exp["obs"], deterministic=True)
    d_act = float(np.abs(act - exp["actions"]).max())
    with th.no_grad():
        v_r, v_c, lp, _ = model.policy.evaluate_actions(th.as_tensor(exp["obs"]).float(), th.as_tensor(exp["actions"]))
    d_vr = float(np.abs(v_r.numpy().ravel() - exp["v_r"]).max()); d_vc = float(np.abs(v_c.numpy().ravel() - exp["v_c"]).max())
    d_lp = float(np.abs(lp.numpy() - exp["log_prob"]).max())
    # hyper-parameters the reference restored from `data`
    assert model.n_steps == 64 and model.batch_size == 64 and model.n_epochs == 2 and model.target_kl == 0.01 and model.learning_rate == 1e-3
    assert abs(model.clip_range(1.0) - 0.2) < 1e-12 and model.algo_type == "lagrangian" and model.n_envs == 4
    # the optimizer state went into the reference's torch Adam (state of every parameter, step counts)
    st = model.policy.optimizer.state_dict()["state"]
    assert len(st) == len(list(model.policy.parameters())) and all(int(s["step"]) > 0 for s in st.values())
    ok = d_act <= 1e-5 and d_vr <= 2e-5 and d_vc <= 2e-5 and d_lp <= 1e-4
    print(("VERIFIED" if ok else "MISMATCH") + f": reference PPOLagrangian.load({archive}) -> predict on 64 observations: max |d action| {d_act:.2e}, "
          f"|d v_r| {d_vr:.2e}, |d v_c| {d_vc:.2e}, |d log_prob| {d_lp:.2e} vs the HIP path; optimizer steps {sorted(set(int(s['step']) for s in st.values()))}")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
