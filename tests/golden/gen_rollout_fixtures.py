"""Golden vectors for the rollout-side fusion from the reference's own rsl_rl classes: a PPO rollout
(rsl_rl/algorithms/ppo.py:94-118 act / process_env_step with the time-out bootstrap) filling a RolloutStorage
(rsl_rl/storage/rollout_storage.py:89-102) and compute_returns (:124-138: GAE + advantage normalisation).

Build-container only:   PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_rollout_fixtures.py
Output: tests/golden/rollout_gae.npz (inputs per step + every storage tensor afterwards)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

rh.load_reference()
import torch  # noqa: E402


def main(N=48, T=24, seed=5):
    from rsl_rl.algorithms import PPO
    from rsl_rl.modules import ActorCritic
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    ac = ActorCritic(45, 61, 12, actor_hidden_dims=[32, 16], critic_hidden_dims=[32, 16], activation="elu", init_noise_std=1.0)
    alg = PPO(ac, device="cpu")
    alg.init_storage(N, T, [45], [61], [12])
    out = {k: [] for k in ("obs", "critic_obs", "actions", "values", "logp", "mu", "sigma", "rew", "dones", "time_outs")}
    obs = torch.from_numpy(rng.normal(size=(N, 45)).astype(np.float32))
    cobs = torch.from_numpy(rng.normal(size=(N, 61)).astype(np.float32))
    with torch.inference_mode():
        for t in range(T):
            actions = alg.act(obs, cobs)
            tr = alg.transition
            out["obs"].append(obs.numpy().copy()); out["critic_obs"].append(cobs.numpy().copy())
            out["actions"].append(actions.numpy().copy()); out["values"].append(tr.values.numpy().copy())
            out["logp"].append(tr.actions_log_prob.numpy().copy()); out["mu"].append(tr.action_mean.numpy().copy())
            out["sigma"].append(tr.action_sigma.numpy().copy())
            rew = torch.from_numpy((rng.normal(size=N) * 0.05 + 0.02).astype(np.float32))
            dones = torch.from_numpy(rng.random(N) < 0.08)
            time_outs = dones & torch.from_numpy(rng.random(N) < 0.5)
            out["rew"].append(rew.numpy().copy()); out["dones"].append(dones.numpy().astype(np.uint8))
            out["time_outs"].append(time_outs.numpy().astype(np.uint8))
            alg.process_env_step(rew, dones, {"time_outs": time_outs})
            obs = torch.from_numpy(rng.normal(size=(N, 45)).astype(np.float32))
            cobs = torch.from_numpy(rng.normal(size=(N, 61)).astype(np.float32))
        last_values = alg.actor_critic.evaluate(cobs).detach()
        alg.compute_returns(cobs)
    st = alg.storage
    arrays = {k: np.stack(v) for k, v in out.items()}
    arrays.update(gamma=np.float32(alg.gamma), lam=np.float32(alg.lam), last_values=last_values.numpy().copy())
    # rows copied verbatim are checked here and not stored twice
    for k, src in (("observations", "obs"), ("privileged_observations", "critic_obs"), ("actions", "actions"), ("mu", "mu"), ("sigma", "sigma")):
        assert np.array_equal(getattr(st, k).numpy(), arrays[src]), k
    assert np.array_equal(st.values.numpy(), arrays["values"]) and np.array_equal(st.actions_log_prob.numpy()[..., 0], arrays["logp"])
    for k in ("rewards", "dones", "returns", "advantages"):
        arrays["st_" + k] = getattr(st, k).numpy().copy()
    path = os.path.join(HERE, "rollout_gae.npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path), "dones", int(arrays["dones"].sum()), "time_outs", int(arrays["time_outs"].sum()),
          "adv mean/std", float(st.advantages.mean()), float(st.advantages.std()))


if __name__ == "__main__":
    main()
