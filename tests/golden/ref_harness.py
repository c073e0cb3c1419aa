"""Loads the reference's PyTorch env stack (read-only /root/reference) against a duck-typed fake
simulator so that golden vectors for the MDP part of the hot path can be generated.

BUILD-CONTAINER ONLY: /root/reference does not exist on the GPU box; the generated .npz files
under tests/golden/ are what travels.  Nothing here is imported by the product or by the tests
that run on the GPU box.

Third-party modules the reference imports but that are absent offline (genesis, cv2, trimesh,
wandb, tensorboard, xlsxwriter, pygame) are registered as empty stub modules; none of their
functionality is exercised on the code paths used.
"""
import inspect
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ["SIMULATOR"] = "genesis"
REF = os.environ.get("LG_REFERENCE", "/root/reference")


class _Anything:
    """Placeholder for names looked up on a stubbed third-party module (type annotations etc.)."""
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, k):
        return _Anything()


class _StubModule(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Anything


def _stub(name):
    m = _StubModule(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


def load_reference():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for n in ("genesis", "cv2", "trimesh", "wandb", "tensorboard", "xlsxwriter", "pygame"):
        if n not in sys.modules:
            try:
                __import__(n)
            except Exception:
                _stub(n)
    if "torch.utils.tensorboard" not in sys.modules:
        try:
            import torch.utils.tensorboard  # noqa: F401
        except Exception:
            tb = _stub("torch.utils.tensorboard")
            tb.SummaryWriter = object
    import legged_gym  # noqa: F401
    return legged_gym


class DrawRecorder:
    """Replacement for math_utils.torch_rand_float / torch.rand_like in the reference modules:
    draws uniforms from a numpy stream and remembers (caller, env_ids, values) per call."""

    def __init__(self, seed):
        import numpy as np
        self.rng = np.random.default_rng(seed)
        self.calls = []

    def _u(self, shape):
        import numpy as np
        import torch
        return torch.from_numpy(self.rng.random(size=tuple(shape), dtype=np.float32))

    def rand_float(self, lower, upper, shape, device):
        st = inspect.stack()
        caller, parent = st[1].function, st[2].function
        env_ids = st[1].frame.f_locals.get("env_ids")
        u = self._u(shape)
        self.calls.append(dict(caller=caller, parent=parent, env_ids=None if env_ids is None else env_ids.clone(), u=u.clone()))
        return (upper - lower) * u + lower

    def rand_like(self, t):
        u = self._u(t.shape)
        self.calls.append(dict(caller="rand_like", parent="", env_ids=None, u=u.clone()))
        return u

    def randint(self, low, high, size, **kw):
        import torch
        st = inspect.stack()
        caller, parent = st[1].function, st[2].function
        env_ids = st[1].frame.f_locals.get("env_ids")
        idx = int(self.rng.integers(low, high))
        self.calls.append(dict(caller="randint:" + caller, parent=parent, env_ids=None if env_ids is None else env_ids.clone(),
                               u=torch.tensor([(idx + 0.5) / high], dtype=torch.float32)))
        return torch.full(tuple(size), idx, dtype=torch.long)

    def tagged(self, tag, shape, env_ids):
        """Uniform draw made by the fake simulator itself (domain randomisation), labelled with its slot name."""
        u = self._u(shape)
        self.calls.append(dict(caller="tag:" + tag, parent="", env_ids=env_ids.clone(), u=u.clone()))
        return u

    def torch_rand(self, *shape, **kw):
        st = inspect.stack()
        caller = st[1].function
        env_ids = st[1].frame.f_locals.get("env_ids")
        u = self._u(shape)
        self.calls.append(dict(caller="torch_rand:" + caller, parent="", env_ids=None if env_ids is None else env_ids.clone(), u=u.clone()))
        return u

    def np_random(self):
        st = inspect.stack()
        env_ids = st[1].frame.f_locals.get("env_ids")
        import torch
        v = float(self.rng.random())
        self.calls.append(dict(caller="np_random:" + st[1].function, parent="", env_ids=None if env_ids is None else env_ids.clone(),
                               u=torch.tensor([v], dtype=torch.float32)))
        return v

    def randint_like(self, t, high, env_ids):
        import numpy as np
        import torch
        idx = self.rng.integers(0, high, size=tuple(t.shape))
        self.calls.append(dict(caller="randint_like", parent="", env_ids=env_ids.clone(),
                               u=torch.from_numpy(((idx + 0.5) / high).astype(np.float32))))
        return torch.from_numpy(idx).to(t.dtype)

    def take(self):
        c, self.calls = self.calls, []
        return c
