"""Rollout-side fusion: the record and GAE kernels behind hcr_genesis_lr_cl_amd.rollout.RolloutStorage against
golden vectors produced by rsl_rl's own PPO + RolloutStorage (tests/golden/rollout_gae.npz).  Tolerance 1e-5 (f32 recurrence
over 24 steps with FMA contraction; mean / std accumulated in f64 on the device, f32 in torch)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rollout_gae.npz")


def test_rollout_oracle_reproduces_rsl_rl():
    from oracle import rollout_oracle as ro
    fx = np.load(GOLD)
    T = fx["rew"].shape[0]
    for t in range(T):
        np.testing.assert_allclose(ro.bootstrap_rewards(fx["rew"][t], fx["values"][t], fx["time_outs"][t], fx["gamma"]), fx["st_rewards"][t][:, 0],
                                   rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(fx["st_dones"][..., 0], fx["dones"])
    ret, adv = ro.compute_returns(fx["values"], fx["st_rewards"], fx["st_dones"], fx["last_values"], fx["gamma"], fx["lam"])
    np.testing.assert_allclose(ret, fx["st_returns"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(adv, fx["st_advantages"], rtol=1e-5, atol=2e-6)
    assert fx["time_outs"].sum() > 10 and fx["dones"].sum() > fx["time_outs"].sum()


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["add_step", "add_transitions"])
def test_rollout_storage_matches_rsl_rl(path):
    """Both ways of filling it: the fused `add_step` (bootstrap in the kernel) and the reference's own entry point
    `add_transitions` (rewards already bootstrapped by PPO.process_env_step)."""
    import torch
    from hcr_genesis_lr_cl_amd.rollout import RolloutStorage
    fx = np.load(GOLD)
    T, N = fx["rew"].shape
    dev = "cuda:0"
    st = RolloutStorage(N, T, [45], [61], [12], dev)
    g = lambda k, t: torch.from_numpy(fx[k][t]).to(dev)
    wide = torch.zeros(N, 200, device=dev)                      # the actor observation arrives as a strided window of a wider row
    for t in range(T):
        wide[:, 100:145] = g("obs", t)
        obs = wide[:, 100:145]
        assert not obs.is_contiguous()
        if path == "add_step":
            st.actions[t].copy_(g("actions", t)); st.values[t].copy_(g("values", t)); st.actions_log_prob[t, :, 0].copy_(g("logp", t))
            st.mu[t].copy_(g("mu", t)); st.sigma[t].copy_(g("sigma", t))
            st.add_step(g("rew", t), g("dones", t).bool(), g("time_outs", t).bool(), float(fx["gamma"]), observations=obs, critic_observations=g("critic_obs", t))
        else:
            tr = RolloutStorage.Transition()
            tr.observations, tr.critic_observations, tr.actions, tr.values = obs, g("critic_obs", t), g("actions", t), g("values", t)
            tr.actions_log_prob, tr.action_mean, tr.action_sigma = g("logp", t), g("mu", t), g("sigma", t)
            tr.rewards = torch.from_numpy(fx["st_rewards"][t][:, 0]).to(dev)          # as PPO.process_env_step leaves them
            tr.dones = g("dones", t).bool()
            st.add_transitions(tr)
    with pytest.raises(AssertionError):
        st.add_step(g("rew", 0), g("dones", 0).bool(), None, 0.0)
    st.compute_returns(torch.from_numpy(fx["last_values"]).to(dev), float(fx["gamma"]), float(fx["lam"]))
    torch.cuda.synchronize()
    c = lambda x: x.cpu().numpy()
    np.testing.assert_array_equal(c(st.observations), fx["obs"]); np.testing.assert_array_equal(c(st.privileged_observations), fx["critic_obs"])
    np.testing.assert_array_equal(c(st.dones), fx["st_dones"])
    np.testing.assert_allclose(c(st.rewards), fx["st_rewards"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(c(st.returns), fx["st_returns"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(c(st.advantages), fx["st_advantages"], rtol=1e-5, atol=5e-6)
    assert abs(float(st.advantages.mean())) < 1e-6 and abs(float(st.advantages.std()) - 1.0) < 1e-5
    batches = list(st.mini_batch_generator(4, 1))
    assert len(batches) == 4 and batches[0][0].shape == (T * N // 4, 45) and batches[0][1].shape == (T * N // 4, 61)
    st.clear()
    assert st.step == 0


@pytest.mark.gpu
def test_zero_copy_observation_rows_follow_the_env():
    """With cfg.hip.obs_sets = T + 1 the storage's observation rows are the env's observation copies: after T steps row t
    holds the observation the policy acted on at step t, without any copy; the next rollout starts from row 0 again."""
    import torch
    from hcr_genesis_lr_cl_amd.config import GO2Cfg
    from hcr_genesis_lr_cl_amd.envs import GO2, set_seed
    from hcr_genesis_lr_cl_amd.rollout import RolloutStorage
    T, N = 6, 64
    cfg = GO2Cfg()
    cfg.env.num_envs = N
    cfg.hip.obs_sets = T + 1
    set_seed(1)
    env = GO2(cfg, None, "cuda:0", True)
    obs, _ = env.reset()
    st = RolloutStorage(N, T, [45], [None], [12], "cuda:0", env=env)
    assert st.zero_copy and st.observations.data_ptr() == env._engine.buf.raw("obs_buf").data_ptr()
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for r in range(3):
        seen = []
        obs = env.get_observations()
        for t in range(T):
            assert obs.data_ptr() == st.observations[t].data_ptr()          # the policy reads the storage row itself
            seen.append(obs.clone())
            st.values[t].zero_()
            obs, _, rew, done, extras = env.step(torch.randn(N, 12, generator=g, device="cuda"))
            st.add_step(rew, done, extras["time_outs"], 0.99)
        torch.cuda.synchronize()
        for t in range(T):
            assert torch.equal(st.observations[t], seen[t]), (r, t)
        last = obs.clone()
        st.clear()
        assert torch.equal(env.get_observations(), last) and env.get_observations().data_ptr() == st.observations[0].data_ptr()
