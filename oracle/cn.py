"""Oracle: constraint-net (zeta_theta) backward step — importance weights, KL early stop, loss, Adam.
Test infrastructure only.

ref: icrl/constraint_net.py:137-229   (ConstraintNet.train)
     icrl/constraint_net.py:231-256   (compute_is_weights)
     icrl/constraint_net.py:301-321   (get, _update_learning_rate)

Quirks kept on purpose:
  * per-step IS weights have shape [N,1]; indexing + ``[..., None]`` gives [B,1,1], which broadcasts
    against log(zeta_N) [B,1] to [B,B,1]: nominal_loss == mean(w) * mean(log zeta_N);
  * per-episode products are taken in float32 and overflow to inf/nan for long episodes; comparisons
    against nan are False so no early stop happens then;
  * with the default full batch (batch_size None) every iteration is one optimiser step.
"""
from itertools import accumulate

import numpy as np
import torch as th


def is_weights_and_kls(preds_old, preds_new, episode_lengths, eps=1e-5, per_step=False):
    """ref: constraint_net.py:231-256.  preds_*: [N,1] float32 tensors.
    Returns (weights, kl_old_new, kl_new_old); weights is [N,1] (per-step) or [N] (per-episode)."""
    with th.no_grad():
        n_ep = len(episode_lengths)
        bounds = [0] + list(accumulate(int(l) for l in episode_lengths))
        ratio = (preds_new + eps) / (preds_old + eps)
        prod = th.tensor([th.prod(ratio[bounds[j]:bounds[j + 1]]) for j in range(n_ep)])
        normed = n_ep * prod / (th.sum(prod) + eps)
        if per_step:
            w = (ratio / th.mean(ratio)).clone()
        else:
            parts = []
            for length, weight in zip(episode_lengths, normed):
                parts += [weight] * int(length)
            w = th.tensor(parts)
        kl_old_new = th.mean(-th.log(prod + eps))
        pm = th.mean(prod)
        kl_new_old = th.mean((prod - pm) * th.log(prod + eps) / (pm + eps))
    return w, kl_old_new, kl_new_old


def cn_loss(net, nominal_batch, expert_batch, is_batch, reg_coeff, eps=1e-5, gail=False, factored=False):
    """ref: constraint_net.py:188-202."""
    nominal_preds = net.forward(nominal_batch)
    expert_preds = net.forward(expert_batch)
    if gail:
        bce = th.nn.BCELoss()
        nominal_loss = bce(nominal_preds, th.zeros(*nominal_preds.size()))
        expert_loss = bce(expert_preds, th.ones(*expert_preds.size()))
        reg = th.tensor(0)
        loss = nominal_loss + expert_loss
    else:
        expert_loss = th.mean(th.log(expert_preds + eps))
        log_nom = th.log(nominal_preds + eps)
        if factored and is_batch.dim() == 3:
            nominal_loss = th.mean(is_batch) * th.mean(log_nom)       # == mean over the [B,B,1] broadcast
        else:
            nominal_loss = th.mean(is_batch * log_nom)
        reg = reg_coeff * (th.mean(1 - expert_preds) + th.mean(1 - nominal_preds))
        loss = (-expert_loss + nominal_loss) + reg
    return loss, expert_loss, nominal_loss, reg, nominal_preds, expert_preds


def cn_train(net, optimizer, iterations, nominal_data, expert_data, episode_lengths, *, reg_coeff=0.0,
             importance_sampling=True, per_step=False, target_kl_old_new=-1, target_kl_new_old=-1,
             eps=1e-5, gail=False, batch_size=None, factored=False, rng=np.random):
    """ref: constraint_net.py:155-229 after prepare_data.  nominal_data / expert_data: [N, d] float32 tensors.
    Returns the ``backward/*`` metrics dict."""
    if importance_sampling:
        with th.no_grad():
            start_preds = net.forward(nominal_data).detach()
    early_stop_itr = iterations
    loss = th.tensor(np.inf)
    for itr in range(iterations):
        if importance_sampling:
            with th.no_grad():
                cur = net.forward(nominal_data).detach()
            w, kl_on, kl_no = is_weights_and_kls(start_preds.clone(), cur.clone(), episode_lengths, eps, per_step)
            if (target_kl_old_new != -1 and kl_on > target_kl_old_new) or \
               (target_kl_new_old != -1 and kl_no > target_kl_new_old):
                early_stop_itr = itr
                break
        else:
            w = th.ones(nominal_data.shape[0])
        n_nom, n_exp = nominal_data.shape[0], expert_data.shape[0]
        if batch_size is None:
            batches = [(np.arange(n_nom), np.arange(n_exp))]
        else:
            size = min(n_nom, n_exp)
            perm = rng.permutation(size)
            batches = [(perm[s:s + batch_size],) * 2 for s in range(0, size, batch_size)]
        for nom_idx, exp_idx in batches:
            is_batch = w[nom_idx][..., None]
            loss, expert_loss, nominal_loss, reg, nominal_preds, expert_preds = cn_loss(
                net, nominal_data[nom_idx], expert_data[exp_idx], is_batch, reg_coeff, eps, gail, factored)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
    m = {"backward/cn_loss": loss.item(),
         "backward/expert_loss": expert_loss.item(),
         "backward/unweighted_nominal_loss": th.mean(th.log(nominal_preds + eps)).item(),
         "backward/nominal_loss": nominal_loss.item(),
         "backward/regularizer_loss": reg.item(),
         "backward/is_mean": th.mean(w).item(), "backward/is_max": th.max(w).item(), "backward/is_min": th.min(w).item(),
         "backward/nominal_preds_max": th.max(nominal_preds).item(),
         "backward/nominal_preds_min": th.min(nominal_preds).item(),
         "backward/nominal_preds_mean": th.mean(nominal_preds).item(),
         "backward/expert_preds_max": th.max(expert_preds).item(),
         "backward/expert_preds_min": th.min(expert_preds).item(),
         "backward/expert_preds_mean": th.mean(expert_preds).item()}
    if importance_sampling:
        m.update({"backward/kl_old_new": kl_on.item(), "backward/kl_new_old": kl_no.item(),
                  "backward/early_stop_itr": early_stop_itr})
    return m
