"""Validation rollout (src/inference.py:20-87): one greedy episode of the real environment, every
move chosen as the arg-max of the search policy, on the GPU search.

Same call as the reference's `model_simulation`.  The history ring is filled with the first
state and action 0 and then carries the real actions (unlike self-play, whose frames all
store action 0: src/training_data_generator.py:110).  Frames returned by `env.render()` are
written with imageio when that package is present; without it (or without frames) the video
is skipped and the reward is still returned.
"""
from __future__ import annotations

import torch
from torch import Tensor

from muzero_amd.ring_buffer import Frame, FrameRingBuffer, make_history_tensor
from muzero_amd.search.factory import create_mcts
from muzero_amd.search.nodes import Node
from muzero_amd.training_data_generator import _make_actions_tensor


@torch.no_grad()
def model_simulation(env, repr_net, dyn_net, pred_net, inference_simulation_depth: int, mcts_config,
                     device: str = "cpu", human_mode: bool = True, video_path: str = "simulation.mp4") -> float:
    print("Starting model simulation...")
    env.reset()
    if human_mode and hasattr(getattr(env, "env", None), "render_mode"):
        env.env.render_mode = "human"
    state = env.get_state()
    mcts = create_mcts(dynamics_network=dyn_net, prediction_network=pred_net, actions=_make_actions_tensor(env, device),
                       config=mcts_config, device=device)
    running_reward = 0.0
    frames = []
    ring = FrameRingBuffer(repr_net.history_length)
    ring.fill(Frame(state, 0))
    for i in range(inference_simulation_depth):
        frames.append(env.render() if hasattr(env, "render") else None)
        history = make_history_tensor(ring).to(device)
        latent = repr_net.to(device)(history)
        root = Node(latent_state=latent, to_play=env.get_to_play())
        tree_policy, value = mcts.run(root)
        action = int(torch.argmax(Tensor(tree_policy)).item())  # first maximum
        state, reward, done = env.step(action)
        running_reward += reward
        ring.add(Frame(state, action))
        if human_mode:
            print(f"Step {i}: Action: {action}, Reward: {reward}, Value: {float(value)}")
        if done:
            break
    frames = [fr for fr in frames if fr is not None]
    if frames:
        try:
            import imageio
        except ImportError:
            print(f"imageio is not installed: {len(frames)} frames not written to {video_path}")
        else:
            print(f"Saving video to {video_path}...")
            imageio.mimsave(video_path, frames, fps=30, macro_block_size=None, ffmpeg_params=["-s", "600x400"])
    return running_reward
