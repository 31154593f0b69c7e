"""ParticleActor -- the reference's particle policy evaluated on the device.

    reference                                                    here
    ---------------------------------------------------------   -----------------------------------------
    networks.actor_particle(obs_others, v_obs, v_goal, ...)      ParticleActor(weights, n_agents, stage)
      (networks.py:517-538)                                        weights: dict keyed by the TF variable names
    probs = (1-eps) probs + eps/l_action; multinomial            actor.act(env, epsilon) -> actions [E, N]
      (alg_credit.py:119-120)                                      (one launch: forward + mixing + sampling)
    alg.run_actor(local_others, local_self, goals, eps, sess)    ParticleRollout.collect(policy=actor, epsilon=..)
      (alg_credit.py:249-270, called at train_onpolicy.py:313)     step and actor launches alternate inside ONE
                                                                   hipGraph; nothing returns to the host

Weights stay float32 [in][out] as TensorFlow shapes them, so a checkpoint exported with
``{v.name: sess.run(v)}`` loads unchanged (names below; a "Policy_main/" prefix and ":0" suffix are ignored).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import Cm3Error


def _epsilon_args(epsilon):
    """(float for the descriptor, device pointer or 0): a float32 device tensor is read by the launch itself, so a
    captured hipGraph follows the annealed epsilon (train_onpolicy.py:369) without being re-captured."""
    if isinstance(epsilon, torch.Tensor):
        if epsilon.dtype != torch.float32 or epsilon.numel() != 1 or epsilon.device.type != "cuda":
            raise Cm3Error("a device epsilon must be one float32 element on the GPU")
        return 0.0, epsilon.data_ptr()
    return float(epsilon), 0

H1_SELF, H1_OTHERS, H2, N_ACTIONS = 64, 128, 64, 5
PRECISIONS = {"f32": 0, "bf16": 1, "f16x3": 2}       # cm3_actor_particle_desc.precision
_NAMES = {
    "w_self": "actor_branch_self/kernel", "b_self": "actor_branch_self/bias", "w_self_h2": "W_branch_self_h2",
    "w_others": "stage-2/actor_others/kernel", "b_others": "stage-2/actor_others/bias",
    "w_others_h2": "stage-2/W_others_h2", "b_h2": "b", "w_out": "actor_out/kernel", "b_out": "actor_out/bias"}


def _canon(name):
    name = name.split(":")[0]
    for prefix in ("Policy_main/", "Policy_target/"):
        if name.startswith(prefix):
            name = name[len(prefix):]
    return name


class ParticleActor(object):
    def __init__(self, weights, n_agents, stage=2, device="cuda:0", seed=12341, env_id_base=0, precision="f32"):
        """precision of the 192 -> 64 second layer (87 % of the network's FLOPs):
        "f32"     exact float32 MFMA -- a k-ordered fmaf chain, the numerics of a plain float32 loop;
        "f16x3"   split float16: activations and weights as hi + lo float16 pairs, three f16 MFMA passes (hi hi + hi lo +
                  lo hi), float32 accumulation.  22 of the 24 significand bits per factor: probabilities within the same 2e-5
                  of the float64 oracle as "f32" (tests/test_gpu_actor.py), ~5x fewer matrix-core cycles.  First-layer
                  activations must stay below float16's 65504 (three orders of magnitude above a trained policy's).  With 5..9
                  agents (16..32 others inputs) actor_others runs in the same split float16 (csrc/actor.hip ActorFirstB::kF16Oth);
        "bf16"    plain bf16 operands: fastest, probabilities within ~1e-2 -- not a parity path."""
        self.device = _lib.require_gpu(device)
        if precision not in PRECISIONS:
            raise Cm3Error("precision must be one of %s" % sorted(PRECISIONS))
        self.precision = precision
        self.n = int(n_agents)
        self.stage = int(stage)
        self.L = 4 * max(self.n - 1, 1)
        self.seed = int(seed)
        self.env_id_base = int(env_id_base)
        src = {_canon(k): v for k, v in weights.items()}
        shapes = {"w_self": (6, H1_SELF), "b_self": (H1_SELF,), "w_self_h2": (H1_SELF, H2), "b_h2": (H2,),
                  "w_out": (H2, N_ACTIONS), "b_out": (N_ACTIONS,)}
        if self.stage > 1:
            shapes.update({"w_others": (self.L, H1_OTHERS), "b_others": (H1_OTHERS,), "w_others_h2": (H1_OTHERS, H2)})
        self.w = {}
        for short, shape in shapes.items():
            name = _NAMES[short]
            if name not in src:
                raise Cm3Error("missing actor weight %r" % name)
            t = torch.as_tensor(np.asarray(src[name]), dtype=torch.float32).contiguous()
            if tuple(t.shape) != shape:
                raise Cm3Error("actor weight %r has shape %s, expected %s" % (name, tuple(t.shape), shape))
            self.w[short] = t.to(self.device)
        self._wt = _lib.ActorParticleWeights()
        for short in _NAMES:
            setattr(self._wt, short, _lib.ptr(self.w.get(short)))
        self._lib = _lib.lib()
        nbytes = self._lib.cm3_actor_particle_packed_bytes(self.n)
        self._packed = torch.zeros(nbytes // 4, dtype=torch.float32, device=self.device)
        self._wt.packed = self._packed.data_ptr()
        self.repack()

    def repack(self):
        """Re-arrange the (possibly updated in place) TF-shaped weights into the forward kernel's layout."""
        d = self._desc(1, 0.0, 0)
        _lib.check(self._lib.cm3_actor_particle_pack(ctypes.byref(d), ctypes.byref(self._wt), self._packed.data_ptr(),
                                                     _lib.current_stream_handle(self.device)))

    def _desc(self, n_envs, epsilon, env_id_base):
        d = _lib.ActorParticleDesc()
        d.n_envs, d.n_agents, d.stage = int(n_envs), self.n, self.stage
        d.n_h1_self, d.n_h1_others, d.n_h2, d.n_actions = H1_SELF, H1_OTHERS, H2, N_ACTIONS
        d.epsilon = float(epsilon)
        d.precision = PRECISIONS[self.precision]
        d.env_id_base = int(env_id_base)
        d.seed = self.seed & 0xFFFFFFFFFFFFFFFF
        return d

    def enqueue(self, n_envs, obs_others, state, goals, meta, episode, actions, epsilon, probs=None, stream=None,
                env_id_base=None):
        """Raw launch on device pointers/tensors (float32 env buffers)."""
        b = _lib.ActorParticleBufs()
        b.obs_others, b.state, b.goals = _lib.ptr(obs_others), _lib.ptr(state), _lib.ptr(goals)
        b.meta, b.episode, b.actions, b.probs = _lib.ptr(meta), _lib.ptr(episode), _lib.ptr(actions), _lib.ptr(probs)
        epsilon, b.epsilon_dev = _epsilon_args(epsilon)
        d = self._desc(n_envs, epsilon, self.env_id_base if env_id_base is None else env_id_base)
        s = _lib.current_stream_handle(self.device) if stream is None else stream
        _lib.check(self._lib.cm3_actor_particle_f32(ctypes.byref(d), ctypes.byref(self._wt), ctypes.byref(b), s))

    def act(self, env, epsilon, return_probs=False):
        """Actions [E, N] int32 for the env's CURRENT observation (alg.run_actor); optionally the mixed
        probabilities [E, N, 5]."""
        if env.dtype != torch.float32:
            raise Cm3Error("the device actor reads float32 env buffers")
        if env.n != self.n:
            raise Cm3Error("actor built for %d agents, env has %d" % (self.n, env.n))
        cur = env._cur
        actions = torch.empty(env.E, env.n, dtype=torch.int32, device=self.device)
        probs = torch.empty(env.E, env.n, N_ACTIONS, dtype=torch.float32, device=self.device) if return_probs else None
        self.enqueue(env.E, env._obs_others[cur], env._state[cur], env._goals, env._meta, env._episode, actions,
                     epsilon, probs, env_id_base=env.env_id_base)
        return (actions, probs) if return_probs else actions


_CK_NAMES = {
    "conv_w": "conv/Conv/weights", "conv_b": "conv/Conv/biases", "lin_w": "conv_linear/kernel",
    "lin_b": "conv_linear/bias", "self_w": "branch_self/kernel", "self_b": "branch_self/bias", "w_self_h2": "W_self_h2",
    "others_w": "stage-2/branch_others/kernel", "others_b": "stage-2/branch_others/bias",
    "w_others_h2": "stage-2/W_others_h2", "b_h2": "b", "out_w": "actor_out/kernel", "out_b": "actor_out/bias"}
CK_CONV_F, CK_CONV_LIN, CK_H1, CK_H2 = 6, 32, 256, 256


class CheckersActor(object):
    """The reference's Checkers policy evaluated on the device.

        reference                                                        here
        -------------------------------------------------------------   ----------------------------------------
        networks.actor_checkers(a_prev, t_obs_self, v_obs_self,          CheckersActor(weights, n_agents, stage)
            v_obs_others, v_goal, f1=6, k1=[3,3], n_h1=256, n_h2=256)      weights: dict keyed by the TF variable names
            (networks.py:549-578; conv: convnet_1 :67-75)
        probs = (1-eps) probs + eps/5; multinomial                       actor.act(env, epsilon, actions_prev)
            (alg_credit_checkers.py:112-113)                               -> actions [E, N] (one launch)
        alg.run_actor(actions_prev, obs_others, obs_self_t, obs_self_v,  CheckersRollout.collect(goals, policy=actor,
            goals, eps, sess)  (alg_credit_checkers.py:229-253)            epsilon=..): actor and step launches alternate
                                                                           inside ONE hipGraph
    """

    def __init__(self, weights, n_agents, stage=2, device="cuda:0", seed=12341, env_id_base=0, precision="f32"):
        """precision: "f32" (default: every layer on the exact-f32 MFMA), "f16x3" (every layer in split float16: activations and
        weights as float16 hi + lo, three float16 MFMAs per product with float32 accumulation -- held to the same 2e-5 parity
        bound as "f32", half its time per launch), or "bf16" (the two 256x256 layers rounded to bf16; not a parity path:
        probabilities within ~1e-2 of the float32 ones)."""
        self.device = _lib.require_gpu(device)
        if precision not in PRECISIONS:
            raise Cm3Error("precision must be one of %s" % sorted(PRECISIONS))
        self.precision = precision
        self.n = int(n_agents)
        self.stage = int(stage)
        self.Lo = 2 * max(self.n - 1, 1)
        self.seed = int(seed)
        self.env_id_base = int(env_id_base)
        src = {_canon(k): v for k, v in weights.items()}
        cat = CK_CONV_LIN + 4 + N_ACTIONS + 2
        shapes = {"conv_w": (3, 3, 3, CK_CONV_F), "conv_b": (CK_CONV_F,), "lin_w": (25 * CK_CONV_F, CK_CONV_LIN),
                  "lin_b": (CK_CONV_LIN,), "self_w": (cat, CK_H1), "self_b": (CK_H1,), "w_self_h2": (CK_H1, CK_H2),
                  "b_h2": (CK_H2,), "out_w": (CK_H2, N_ACTIONS), "out_b": (N_ACTIONS,)}
        if self.stage > 1:
            shapes.update({"others_w": (self.Lo, CK_H1), "others_b": (CK_H1,), "w_others_h2": (CK_H1, CK_H2)})
        self.w = {}
        for short, shape in shapes.items():
            name = _CK_NAMES[short]
            if name not in src:
                raise Cm3Error("missing actor weight %r" % name)
            t = torch.as_tensor(np.asarray(src[name]), dtype=torch.float32).contiguous()
            if tuple(t.shape) != shape:
                raise Cm3Error("actor weight %r has shape %s, expected %s" % (name, tuple(t.shape), shape))
            self.w[short] = t.to(self.device)
        self._wt = _lib.ActorCheckersWeights()
        for short in _CK_NAMES:
            setattr(self._wt, short, _lib.ptr(self.w.get(short)))
        self._lib = _lib.lib()
        nbytes = self._lib.cm3_actor_checkers_packed_bytes()
        self._packed = torch.zeros(nbytes // 4, dtype=torch.float32, device=self.device)
        self._wt.packed = self._packed.data_ptr()
        self.repack()

    def _desc(self, n_envs, epsilon, env_id_base, obst_stride):
        d = _lib.ActorCheckersDesc()
        d.n_envs, d.n_agents, d.stage, d.n_obs = int(n_envs), self.n, self.stage, 2
        d.conv_f, d.n_conv_linear, d.n_h1, d.n_h2, d.n_actions = CK_CONV_F, CK_CONV_LIN, CK_H1, CK_H2, N_ACTIONS
        d.epsilon = float(epsilon)
        d.precision = PRECISIONS[self.precision]
        d.obs_self_t_stride = int(obst_stride)
        d.env_id_base = int(env_id_base)
        d.seed = self.seed & 0xFFFFFFFFFFFFFFFF
        return d

    def repack(self):
        d = self._desc(1, 0.0, 0, 75 * self.n)
        _lib.check(self._lib.cm3_actor_checkers_pack(ctypes.byref(d), ctypes.byref(self._wt), self._packed.data_ptr(),
                                                     _lib.current_stream_handle(self.device)))

    def enqueue(self, n_envs, obs_self_t_raw, obst_stride, obs_self_v, obs_others, goals, actions_prev, steps, episode,
                actions, epsilon, probs=None, stream=None, env_id_base=None, prev_done=None):
        """Raw launch on the env's device buffers (obs_self_t_raw: the int8 storage with obst_stride bytes per env).
        prev_done (uint8 [E], optional): envs that finished an episode on the previous tick see actions_prev = 0."""
        b = _lib.ActorCheckersBufs()
        b.obs_self_t, b.obs_self_v, b.obs_others = _lib.ptr(obs_self_t_raw), _lib.ptr(obs_self_v), _lib.ptr(obs_others)
        b.goals, b.actions_prev, b.steps, b.episode = (_lib.ptr(goals), _lib.ptr(actions_prev), _lib.ptr(steps),
                                                       _lib.ptr(episode))
        b.actions, b.probs = _lib.ptr(actions), _lib.ptr(probs)
        b.prev_done = _lib.ptr(prev_done)
        epsilon, b.epsilon_dev = _epsilon_args(epsilon)
        d = self._desc(n_envs, epsilon, self.env_id_base if env_id_base is None else env_id_base, obst_stride)
        s = _lib.current_stream_handle(self.device) if stream is None else stream
        _lib.check(self._lib.cm3_actor_checkers_f32(ctypes.byref(d), ctypes.byref(self._wt), ctypes.byref(b), s))

    def enqueue_rollout(self, env_desc, traj, n_envs, obst_stride, n_ticks, epsilon, prev0=None, probs=None, stream=None,
                        final_obs=None, prev0_next=None):
        """The whole policy-driven rollout in ONE launch (cm3_policy_rollout_checkers): env_desc / traj are the env's descriptor and
        the rollout's trajectory struct; probs: optional float32 [T, E, N, 5]; final_obs: optional _lib.CheckersBufs naming the env's
        current-observation buffers, which then receive the state the rollout leaves."""
        epsilon, eps_dev = _epsilon_args(epsilon)
        d = self._desc(n_envs, epsilon, env_desc.env_id_base, obst_stride)
        s = _lib.current_stream_handle(self.device) if stream is None else stream
        _lib.check(self._lib.cm3_policy_rollout_checkers(
            ctypes.byref(env_desc), ctypes.byref(traj), ctypes.byref(d), ctypes.byref(self._wt), _lib.ptr(prev0), _lib.ptr(prev0_next), _lib.ptr(probs),
            0 if probs is None else probs[0].numel() * probs.element_size(), eps_dev,
            None if final_obs is None else ctypes.byref(final_obs), int(n_ticks), s))

    def fused_rollout_ok(self, env):
        """cm3_policy_rollout_checkers covers the reference's configurations: split-float16 actor, one agent at stage 1 or two at
        stage 2, the 3 x 8 band with n_obs 2 -- and actor and env must share seed and env_id_base (one Philox key; otherwise
        collect() falls back to a launch pair per tick, whose actor launch takes its own key)."""
        same_key = (self.seed & 0xFFFFFFFFFFFFFFFF) == int(env._desc.seed) and self.env_id_base == int(env._desc.env_id_base)
        return (self.precision == "f16x3" and same_key and env.n == self.n and env.n in (1, 2) and (self.stage > 1) == (env.n > 1)
                and env.K == 5 and env.R == 3 and env.C == 8 and env.grid_stride % 4 == 0 and env.obst_stride % 4 == 0)

    def act(self, env, epsilon, actions_prev=None, return_probs=False):
        """Actions [E, N] int32 for the env's CURRENT observation (alg.run_actor); actions_prev None = zeros
        (train_onpolicy.py:295)."""
        if env.n != self.n:
            raise Cm3Error("actor built for %d agents, env has %d" % (self.n, env.n))
        if env.K != 5:
            raise Cm3Error("the device actor reads 5x5 windows (n_obs = 2)")
        s = env._slots[env._cur]
        prev = None
        if actions_prev is not None:
            prev = torch.as_tensor(actions_prev, device=self.device).to(torch.int32).reshape(env.E, env.n).contiguous()
        actions = torch.empty(env.E, env.n, dtype=torch.int32, device=self.device)
        probs = torch.empty(env.E, env.n, N_ACTIONS, dtype=torch.float32, device=self.device) if return_probs else None
        self.enqueue(env.E, s["obs_self_t_raw"], env.obst_stride, s["obs_self_v"], s["obs_others"], env._goals, prev,
                     env._steps, env._episode, actions, epsilon, probs, env_id_base=env._desc.env_id_base)
        return (actions, probs) if return_probs else actions
