"""Batched die-reorient environment — ``CustomReorientEnv`` (/root/reference/src/envs/reorient.py:11-212,
registrations ``CustomMyoChallengeDieReorientP1/P2-v0``, src/envs/__init__.py:24-55).

Physics runs in libmyobatch (``myo_batch_physics_step``: frame_skip = 5 substeps of the hand + die
model per env step, one environment per wavefront); the task layer — observation, reward dictionary,
goal sampling, reward shaping state, TimeLimit, auto-reset — is torch code on the same device (it is
negligible next to the physics).  What is pinned and what is recalled:

* ``get_reward_dict`` (reorient.py:12-56) is checked against goldens produced by calling the reference
  function itself (tools/make_golden.py -> tests/golden/reorient_reward_goldens.npz);
* ``reset`` / ``set_orientation`` (:124-205) follow the reference line by line (goal position = initial +
  U(goal_pos)^3, Euler angles U(range) per axis with the optional ``goal_rot_x/y/z`` range lists, RSI by
  linear interpolation between the default die pose and the goal pose);
* the observation layout is MyoSuite 1.2.3's ``ReorientEnvV0`` [3P-RECALL, no artifact pins it]:
  hand_qpos 23, hand_qvel 23 (x dt), obj_pos 3, goal_pos 3, pos_err 3, obj_rot 3, goal_rot 3, rot_err 3,
  act 39 = 103; Euler angles by the mujoco-py ``rotations.py`` convention MyoSuite copies.

Phase 2's per-episode die randomisation (``obj_size_change``, ``obj_friction_change``, reorient.py:136-147)
goes through ``myo_batch_set_object_group``: per env one size delta (every die geom moves outward by it,
edge capsules grow by it — the reference's geometry update) and ONE friction triple for the die (the
reference draws an independent triple for each of the die's geoms).  The die of the synthetic model is a
rounded cube made of corner spheres and edge capsules (synth_hand.py).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .. import native
from ..model import CompiledModel, compile_model

REGISTRATION = {      # src/envs/__init__.py:24-55
    "CustomMyoReorientP1": dict(max_episode_steps=150, kwargs=dict(normalize_act=True, frame_skip=5, goal_pos=(-0.010, 0.010),
                                                                   goal_rot=(-1.57, 1.57))),
    "CustomMyoReorientP2": dict(max_episode_steps=150, kwargs=dict(normalize_act=True, frame_skip=5, goal_pos=(-0.020, 0.020),
                                                                   goal_rot=(-3.14, 3.14), obj_size_change=0.007,
                                                                   obj_friction_change=(0.2, 0.001, 0.00002))),
}
SETUP_DEFAULTS = dict(   # reorient.py:58-76; DEFAULT_RWD_KEYS_AND_WEIGHTS of ReorientEnvV0 [3P-RECALL]
    weighted_reward_keys={"pos_dist": 100.0, "rot_dist": 1.0}, goal_pos=(0.0, 0.0), goal_rot=(0.785, 0.785), obj_size_change=0,
    obj_friction_change=(0, 0, 0), pos_th=0.025, rot_th=0.262, drop_th=0.200, enable_rsi=False, rsi_distance_pos=0,
    rsi_distance_rot=0, goal_rot_x=None, goal_rot_y=None, goal_rot_z=None, frame_skip=5, normalize_act=True)
RWD_KEYS = ("pos_dist", "rot_dist", "pos_dist_diff", "rot_dist_diff", "alive", "act_reg", "sparse", "solved", "done", "dense")


# ---- rotations (mujoco-py rotations.py convention, as in myosuite.utils.quat_math) [3P-RECALL]
def euler2quat(e: torch.Tensor) -> torch.Tensor:
    ai, aj, ak = e[..., 2] / 2, -e[..., 1] / 2, e[..., 0] / 2
    si, sj, sk = torch.sin(ai), torch.sin(aj), torch.sin(ak)
    ci, cj, ck = torch.cos(ai), torch.cos(aj), torch.cos(ak)
    cc, cs, sc, ss = ci * ck, ci * sk, si * ck, si * sk
    return torch.stack([cj * cc + sj * ss, cj * cs - sj * sc, -(cj * ss + sj * cc), cj * sc - sj * cs], -1)


def quat2mat(q: torch.Tensor) -> torch.Tensor:
    q = q / torch.clamp(torch.linalg.norm(q, dim=-1, keepdim=True), min=1e-30)
    w, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(q.shape[:-1] + (3, 3))


def mat2euler(m: torch.Tensor) -> torch.Tensor:
    cy = torch.sqrt(m[..., 2, 2] ** 2 + m[..., 1, 2] ** 2)
    cond = cy > 1e-12
    ez = torch.where(cond, -torch.atan2(m[..., 0, 1], m[..., 0, 0]), -torch.atan2(-m[..., 1, 0], m[..., 1, 1]))
    ey = -torch.atan2(-m[..., 0, 2], cy)
    ex = torch.where(cond, -torch.atan2(m[..., 1, 2], m[..., 2, 2]), torch.zeros_like(cy))
    return torch.stack([ex, ey, ez], -1)


def get_reward_dict(obs_pos_err, obs_rot_err, act, prev_pos_dist, prev_rot_dist, na, drop_th, pos_th, rot_th, weights):
    """reorient.py:12-56 on batched float64 tensors.  Returns the OrderedDict's entries as a dict of [N] tensors."""
    pos_dist_new = torch.abs(torch.linalg.norm(obs_pos_err, dim=-1))
    rot_dist_new = torch.abs(torch.linalg.norm(obs_rot_err, dim=-1))
    act_mag = torch.linalg.norm(act, dim=-1) / na if na != 0 else torch.zeros_like(pos_dist_new)
    drop = pos_dist_new > drop_th
    d = {
        "pos_dist": -1.0 * pos_dist_new, "rot_dist": -1.0 * rot_dist_new,
        "pos_dist_diff": prev_pos_dist - pos_dist_new, "rot_dist_diff": prev_rot_dist - rot_dist_new,
        "alive": (~drop).to(pos_dist_new.dtype), "act_reg": -1.0 * act_mag, "sparse": -rot_dist_new - 10.0 * pos_dist_new,
        "solved": ((pos_dist_new < pos_th) & (rot_dist_new < rot_th) & (~drop)).to(pos_dist_new.dtype),
        "done": drop.to(pos_dist_new.dtype),
    }
    d["dense"] = sum(float(w) * d[k] for k, w in weights.items())
    return d, pos_dist_new, rot_dist_new


class _Box:
    def __init__(self, lo, hi, n):
        self.low, self.high, self.shape, self.dtype = np.full(n, lo, np.float32), np.full(n, hi, np.float32), (n,), np.float32


class ReorientVecEnv:
    """Tensor API of BaodingVecEnv (reset_tensor / step_tensor -> obs, rew, done, trunc, term_obs, comps, ep_info)."""

    def __init__(self, env_name: str, num_envs: int, config: dict, device: int = 0, seed: int = 0, dtype: str = "mixed",
                 model=None, lib: Optional[native.NativeLib] = None, integrator=None):
        if env_name not in REGISTRATION:
            raise ValueError("Environment name not recognized:", env_name)
        reg = REGISTRATION[env_name]
        p = dict(SETUP_DEFAULTS)
        p.update(reg["kwargs"])
        horizon = config.pop("max_episode_steps", None) if isinstance(config, dict) else None
        for k, v in (config or {}).items():
            if k not in p and k not in ("obs_keys", "model_path"):
                raise TypeError(f"{env_name}: unexpected keyword argument {k!r}")
            p[k] = v
        self.p = p
        self.max_episode_steps = int(reg["max_episode_steps"] if horizon is None else horizon)
        if model is None:
            from ..synth_hand import synthetic_hand_die
            model = synthetic_hand_die()
        if not isinstance(model, CompiledModel):
            integ = None if integrator is None else {"euler": 0, "rk4": 1}[integrator.lower()]
            # the die is a rounded cube of 8 corner spheres + 12 edge capsules; its capsule-vs-palm-BOX pairs have no
            # narrow phase here and are dropped ON PURPOSE (the corner spheres carry the contact with the box)
            model = compile_model(model, integrator=integ, unsupported_contacts="drop")
        self.compiled = model
        self.lib = lib or native.load()
        if self.lib.is_emulation:
            self.device = torch.device("cpu")
        else:
            if not torch.cuda.is_available():
                raise native.MyoError("ReorientVecEnv needs a GPU: libmyobatch has no CPU execution path")
            self.device = torch.device(f"cuda:{device}")
        self._model = native.Model(model, self.lib)
        self.dtype = {"mixed": native.MYO_MIXED, "f32": native.MYO_MIXED, "f64": native.MYO_F64}[dtype]      # "f32": round 1's name of the mixed stepper
        self.batch = native.Batch(self._model, None, num_envs, device, seed, self.dtype)      # physics only
        d, N = self.device, num_envs
        self.num_envs = N
        self.nq, self.nv, self.na, self.act_dim = (self._model.size(k) for k in ("nq", "nv", "na", "nu"))
        self.n_hand = self.nq - 7
        self.obs_dim = 2 * self.n_hand + 18 + self.na
        self.observation_space, self.action_space = _Box(-10.0, 10.0, self.obs_dim), _Box(-1.0, 1.0, self.act_dim)
        self.frame_skip = int(p["frame_skip"])
        self.dt = self.frame_skip * float(model.fields["opt_f64"][0])
        self.object_bid, self.goal_bid = model.name2id("body", "Object"), model.name2id("body", "target")
        f64 = lambda a: torch.as_tensor(np.asarray(a, np.float64), device=d)
        body_pos, body_quat = np.asarray(model.fields["body_pos"]).reshape(-1, 3), np.asarray(model.fields["body_quat"]).reshape(-1, 4)
        self.goal_init_pos = f64(body_pos[self.goal_bid])                         # site target_o sits at the body origin
        self.default_init_pos, self.default_init_rot = f64(body_pos[self.object_bid]), f64(body_quat[self.object_bid])
        self.goal_obj_offset = self.goal_init_pos - self.default_init_pos         # reorient.py:83-86
        self.init_qpos = f64(np.asarray(model.fields["qpos0"]).reshape(-1)).clone()
        self.init_qpos[:-7] = 0                                                   # reorient.py:120-121
        self.init_qpos[0] = -1.5
        gb = np.asarray(model.fields["geom_bodyid"]).reshape(-1)
        die_geoms = np.nonzero(gb == self.object_bid)[0]
        self.object_gid0, self.object_gidn = int(die_geoms[0]), int(die_geoms[-1]) + 1
        assert self.object_gidn - self.object_gid0 == len(die_geoms), "die geoms must be contiguous"
        self.nominal_friction = f64(np.asarray(model.fields["geom_friction"]).reshape(-1, 3)[self.object_gid0])
        self.physical_randomisation_applied = bool(p["obj_size_change"]) or any(float(x) != 0 for x in p["obj_friction_change"])
        if self.physical_randomisation_applied:
            self.batch.set_object_group(self.object_gid0, self.object_gidn)
        self._ball_d = torch.zeros((num_envs, 10), dtype=torch.float64, device=d)
        self._ball_d[:, 2:5] = self.nominal_friction
        self._fric_change = f64(np.asarray(p["obj_friction_change"], np.float64))
        self.sync_free = False        # True: fixed-shape, host-sync-free step (graph capture); see _reset_rows
        self.use_graph = True         # replay the step (physics + task layer) from a hipGraph after the first call
        self._graph, self._act_static = None, None
        self.gen = torch.Generator(device=d)
        self.gen.manual_seed(int(seed) + 7919)
        z = lambda *s, dt=torch.float64: torch.zeros(s, dtype=dt, device=d)
        self.goal_pos, self.goal_quat = z(N, 3), z(N, 4)
        self.pos_dist, self.rot_dist = z(N), z(N)
        self.elapsed, self.ep_len, self.ep_ret = z(N, dt=torch.long), z(N), z(N)
        self._qp, self._qv, self._ac, self._tm = z(N, self.nq), z(N, self.nv), z(N, self.na), z(N)
        self._ctrl = z(N, self.act_dim)
        self._obs = z(N, self.obs_dim, dt=torch.float32)
        self._term = z(N, self.obs_dim, dt=torch.float32)
        self._comps = z(N, native.N_RWD, dt=torch.float32)
        # static output buffers, like BaodingVecEnv (PPO's graph-captured rollout reads them by address)
        self._rew, self._ep = z(N, dt=torch.float32), z(N, 2, dt=torch.float32)
        self._done, self._trunc = z(N, dt=torch.uint8), z(N, dt=torch.uint8)
        self.rwd_dict = {}
        self._closed = False

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return None if self.device.type != "cuda" else torch.cuda.current_stream(self.device).cuda_stream

    def _uniform(self, lo, hi, shape):
        return lo + (hi - lo) * torch.rand(shape, generator=self.gen, device=self.device, dtype=torch.float64)

    def _pull(self):
        self.batch.get_state(self._qp, self._qv, self._ac, self._tm, self._stream())

    def _obs_dict(self):
        qp, qv = self._qp, self._qv
        o = {"hand_qpos": qp[:, :self.n_hand], "hand_qvel": qv[:, :self.n_hand] * self.dt, "obj_pos": qp[:, -7:-4],
             "goal_pos": self.goal_pos}
        o["pos_err"] = o["goal_pos"] - o["obj_pos"] - self.goal_obj_offset
        o["obj_rot"] = mat2euler(quat2mat(qp[:, -4:]))
        o["goal_rot"] = mat2euler(quat2mat(self.goal_quat))
        o["rot_err"] = o["goal_rot"] - o["obj_rot"]
        o["act"] = self._ac
        return o

    @staticmethod
    def _flat(o):
        return torch.cat([o[k] for k in ("hand_qpos", "hand_qvel", "obj_pos", "goal_pos", "pos_err", "obj_rot", "goal_rot", "rot_err",
                                          "act")], -1).to(torch.float32)

    def _axis_range(self, choices, n):
        """set_orientation (:183-205): one (low, high) per env, drawn from the optional list of ranges."""
        if choices is None:
            lo = torch.full((n,), float(self.p["goal_rot"][0]), dtype=torch.float64, device=self.device)
            return lo, torch.full_like(lo, float(self.p["goal_rot"][1]))
        ch = torch.as_tensor(np.asarray(choices, np.float64), device=self.device)
        pick = torch.randint(0, ch.shape[0], (n,), generator=self.gen, device=self.device)
        return ch[pick, 0], ch[pick, 1]

    def _reset_rows(self, mask: torch.Tensor):
        """reset() of the envs selected by mask (:124-181); state buffers must be current (_pull).

        Two forms of the same code: indexed (default: one host read of the mask, work only for the rows that
        reset) and, with ``sync_free``, full width with the mask applied by selects — fixed shapes and no host
        synchronisation, which is what a hipGraph capture of the env step needs (PPO's recurrent rollout)."""
        N = self.num_envs
        if self.sync_free:
            idx, n = None, N
        else:
            idx = mask.nonzero().flatten()
            n = idx.numel()
            if n == 0:
                return

        def put(dst, val):
            if idx is not None:
                dst[idx] = val
            else:
                m = mask.view(-1, *([1] * (dst.dim() - 1)))
                v = val if torch.is_tensor(val) else torch.full_like(dst, val)
                dst.copy_(torch.where(m, v.to(dst.dtype), dst))

        take = (lambda src: src[idx]) if idx is not None else (lambda src: src)
        p = self.p
        put(self.goal_pos, self.goal_init_pos + self._uniform(float(p["goal_pos"][0]), float(p["goal_pos"][1]), (n, 3)))
        e = []
        for choices in (p["goal_rot_x"], p["goal_rot_y"], p["goal_rot_z"]):
            lo, hi = self._axis_range(choices, n)
            e.append(lo + (hi - lo) * torch.rand(n, generator=self.gen, device=self.device, dtype=torch.float64))
        put(self.goal_quat, euler2quat(torch.stack(e, -1)))
        if self.physical_randomisation_applied:          # :136-147 (one friction triple per env; see the module docstring)
            fr = self.nominal_friction + (2 * torch.rand((n, 3), generator=self.gen, device=self.device,
                                                         dtype=torch.float64) - 1) * self._fric_change
            c = float(p["obj_size_change"])
            bd = take(self._ball_d).clone()
            bd[:, 2:5] = fr
            bd[:, 8] = self._uniform(-c, c, (n,))
            put(self._ball_d, bd)
            self.batch.set_task(None, None, self._ball_d, self._stream())
        qpos = self.init_qpos.expand(n, -1).clone()
        if p["enable_rsi"]:        # :150-176: the die starts between its default pose and the goal pose
            a, b = float(p["rsi_distance_pos"]), float(p["rsi_distance_rot"])
            qpos[:, -7:-4] = a * self.default_init_pos + (1 - a) * (take(self.goal_pos) - self.goal_obj_offset)
            q = b * self.default_init_rot + (1 - b) * take(self.goal_quat)
            qpos[:, -4:] = q / torch.clamp(torch.linalg.norm(q, dim=-1, keepdim=True), min=1e-30)
        put(self._qp, qpos)
        put(self._qv, 0.0)
        put(self._ac, 0.0)
        put(self._tm, 0.0)
        self.batch.set_state(self._qp, self._qv, self._ac, self._tm, self._stream())
        put(self.elapsed, 0)
        put(self.ep_len, 0.0)
        put(self.ep_ret, 0.0)
        o = self._obs_dict()
        put(self.pos_dist, torch.abs(torch.linalg.norm(take(o["pos_err"]), dim=-1)))       # :178-179
        put(self.rot_dist, torch.abs(torch.linalg.norm(take(o["rot_err"]), dim=-1)))

    # ------------------------------------------------------------------ tensor API
    @torch.no_grad()
    def reset_tensor(self):
        self._pull()
        self._reset_rows(torch.ones(self.num_envs, dtype=torch.bool, device=self.device))
        self._obs.copy_(self._flat(self._obs_dict()))
        return self._obs

    def _step_core(self, actions):
        """Action map, frame_skip physics substeps, observation / reward / termination and the episode counters:
        fixed shapes, no host synchronisation, every result lands in a static buffer (so the whole sequence —
        the physics kernel plus ~25 small tensor kernels — replays from one hipGraph, see step_tensor)."""
        a = torch.clamp(actions, -1.0, 1.0)
        if self.p["normalize_act"]:                      # BaseV0.step: float32 sigmoid(5(a - 0.5)) for muscles
            a = 1.0 / (1.0 + torch.exp(-5.0 * (a - 0.5)))
        self._ctrl.copy_(a)
        self.batch.physics_step(self._ctrl, self.frame_skip, self._stream())
        self._pull()
        o = self._obs_dict()
        bad = ~(torch.isfinite(self._qp).all(-1) & torch.isfinite(self._qv).all(-1))
        rd, pd, rdist = get_reward_dict(o["pos_err"], o["rot_err"], o["act"], self.pos_dist, self.rot_dist, self.na,
                                        float(self.p["drop_th"]), float(self.p["pos_th"]), float(self.p["rot_th"]),
                                        self.p["weighted_reward_keys"])
        self.rwd_dict = rd
        self.pos_dist.copy_(pd); self.rot_dist.copy_(rdist)                       # step(): :207-212
        rew = torch.where(bad, torch.zeros_like(rd["dense"]), rd["dense"])
        self.elapsed += 1
        self.ep_len += 1
        self.ep_ret += rew
        fall = (rd["done"] > 0) | bad
        trunc = (self.elapsed >= self.max_episode_steps) & ~fall
        done = fall | trunc
        obs = self._flat(o)
        obs = torch.where(bad.unsqueeze(-1), torch.zeros_like(obs), obs)
        self._term.copy_(obs)
        self._obs.copy_(obs)
        comps = torch.stack([rd["pos_dist"], rd["rot_dist"], rd["act_reg"], rd["alive"], rd["sparse"], rd["solved"], rd["done"], rd["dense"]], -1)
        self._comps.copy_(torch.nan_to_num(comps).to(torch.float32))
        self._ep.copy_(torch.stack([self.ep_ret, self.ep_len], -1).to(torch.float32))
        self._rew.copy_(rew); self._done.copy_(done); self._trunc.copy_(trunc)

    @torch.no_grad()
    def step_tensor(self, actions):
        a = actions.to(device=self.device, dtype=torch.float32)
        graphable = self.use_graph and self.device.type == "cuda" and not torch.cuda.is_current_stream_capturing()
        if graphable and self._graph is not None:
            self._act_static.copy_(a)
            self.batch.bind_constants(self._stream())     # another batch (an eval env ...) may have launched since
            self._graph.replay()
        else:
            self._step_core(a)
            if graphable:          # this eager step was the warm-up (constants bound, library handles made): capture the next ones
                self._act_static = a.clone()
                self._graph = torch.cuda.CUDAGraph()
                snap = [t.clone() for t in self._graph_state()]
                with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
                    self._step_core(self._act_static)
                for t, v in zip(self._graph_state(), snap):   # (capture records, it does not execute; restored for safety)
                    t.copy_(v)
        done = self._done.bool()
        if self.sync_free or bool(done.any()):
            self._reset_rows(done)
            self._obs.copy_(torch.where(done.unsqueeze(-1), self._flat(self._obs_dict()), self._obs))
        return self._obs, self._rew, self._done, self._trunc, self._term, self._comps, self._ep

    def _graph_state(self):
        return [self.pos_dist, self.rot_dist, self.elapsed, self.ep_len, self.ep_ret, self._obs, self._term, self._comps, self._ep,
                self._rew, self._done, self._trunc, self._ctrl, self._qp, self._qv, self._ac, self._tm]

    # ------------------------------------------------------------------ numpy protocol (subset)
    def reset(self):
        return self.reset_tensor().cpu().numpy().copy()

    def step(self, actions):
        o, r, d, t, term, comps, ep = self.step_tensor(torch.as_tensor(np.asarray(actions, np.float32), device=self.device))
        dh, th = d.cpu().numpy().astype(bool), t.cpu().numpy().astype(bool)
        infos = [{} if not dh[i] else {"terminal_observation": term[i].cpu().numpy(), "TimeLimit.truncated": bool(th[i]),
                                       "episode": {"r": float(ep[i, 0]), "l": int(ep[i, 1])}} for i in range(self.num_envs)]
        return o.cpu().numpy().copy(), r.cpu().numpy(), dh, infos

    def close(self):
        if not self._closed:
            self.batch.close()
            self._closed = True
