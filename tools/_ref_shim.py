"""Import preamble for running the reference implementation in the build container.

Only used by tools/make_goldens.py (golden-vector generation).  It registers empty
stand-in modules for third-party packages the reference imports at module scope but
that this container lacks (pygame, imageio, wandb, gym, gymnasium, pettingzoo,
flappy_bird_gymnasium), back-fills enum.StrEnum for Python 3.10 and puts the reference
checkout on sys.path.  Nothing here is shipped to, or usable on, the GPU box.
"""
import enum
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("MUZERO_REFERENCE_ROOT", "/root/reference")


def install() -> None:
    sys.dont_write_bytecode = True
    if not hasattr(enum, "StrEnum"):
        class StrEnum(str, enum.Enum):
            def __str__(self) -> str:
                return str(self.value)
        enum.StrEnum = StrEnum

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    stub("pygame")
    stub("imageio", mimsave=lambda *a, **k: None)
    stub("wandb", log=lambda *a, **k: None, init=lambda *a, **k: None, finish=lambda *a, **k: None)
    stub("gym"); stub("gym.envs"); stub("gym.envs.box2d")
    stub("gym.envs.box2d.car_racing", CarRacing=_Dummy)
    stub("gym.envs.box2d.lunar_lander", LunarLander=_Dummy)
    stub("gym.spaces"); stub("gym.spaces.discrete", Discrete=_Dummy)
    stub("gymnasium"); stub("gymnasium.spaces", Box=_Dummy)
    stub("pettingzoo"); stub("pettingzoo.classic", connect_four_v3=_Dummy)
    stub("flappy_bird_gymnasium"); stub("flappy_bird_gymnasium.envs")
    stub("flappy_bird_gymnasium.envs.constants", PIPE_HEIGHT=320)
    stub("flappy_bird_gymnasium.envs.flappy_bird_env", FlappyBirdEnv=_Dummy)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
