"""policy architectures beyond three two-layer branches (stable_baselines3/common/torch_layers.py:129-254, icrl/utils.py:636-655):
a shared trunk, branches of 0..4 layers, a trunk alone — the shapes icrl_policy_t.arch describes (csrc/generic.hip)."""

ARCHES = {"trunk": [48, dict(pi=[64, 32, 32], vf=[40], cvf=[])], "deep": [dict(pi=[32], vf=[64, 64, 64], cvf=[96, 200, 64, 16])],
          "trunk-only": [64, 64], "bare": [dict(pi=[], vf=[24], cvf=[])],
          # the largest architecture the library takes: ~1 M parameters = ~3 900 workgroups of the generic Adam launch, more than are
          # resident at once (the target-KL stop word written by workgroup 0 must not be seen by late workgroups of that launch)
          "huge": [256, 256, 256, 256, dict(pi=[256] * 4, vf=[256] * 4, cvf=[256] * 4)]}


def oracle_arch_kwargs(net_arch):
    """net_arch -> the oracle policy's (hidden, shared) keywords."""
    n_sh = next((i for i, x in enumerate(net_arch) if isinstance(x, dict)), len(net_arch))
    d = net_arch[n_sh] if n_sh < len(net_arch) else {}
    return dict(hidden=dict(policy_net=tuple(d.get("pi", ())), value_net=tuple(d.get("vf", ())), cost_value_net=tuple(d.get("cvf", ()))),
                shared=tuple(net_arch[:n_sh]))
