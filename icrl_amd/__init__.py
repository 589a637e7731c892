"""icrl_amd — MI355X-native rollout+update hot path of Inverse Constrained RL (shehryar-malik/icrl).

Host side mirrors the reference's Python surface (ConstraintNet, VecEnv wrappers, RolloutBufferWithCost,
PPOLagrangian, DualVariable, run_me.py / icrl.py entry points); the arithmetic runs in hand-written gfx950
kernels behind the C ABI of include/icrl_hip.h (icrl_amd/csrc -> icrl_amd/lib/libicrl_hip.so).
"""
__version__ = "0.1.0"
