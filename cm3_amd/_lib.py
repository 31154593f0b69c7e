"""ctypes binding of libcm3_hip.so (the C ABI declared in include/cm3_amd.h).

The reference names a "thin C-ABI/cffi layer"; cffi is not installed in this image, so the
stdlib ``ctypes`` is used.  There is NO CPU fallback: if the HIP library is missing or fails to
load, importing anything that computes raises ``Cm3Error`` loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CM3_AMD_LIB: load another build of the SAME ABI instead (tools/*_ab.py compare two builds on one box)
LIB_PATH = os.environ.get("CM3_AMD_LIB") or os.path.join(_HERE, "libcm3_hip.so")
MAX_AGENTS = 10
ABI_VERSION = 7

FLAG_AUTO_RESET = 1
FLAG_GEN_ACTIONS = 2
FLAG_KERNEL_LANE_PER_ENV = 0x100
FLAG_FUSED_TICKS = 0x400
FLAG_KERNEL_LANE_PER_PAIR = 0x200
FLAG_KERNEL_LANE_PER_AGENT = 0x800
KERNEL_FLAGS = {"auto": 0, "env": FLAG_KERNEL_LANE_PER_ENV, "pair": FLAG_KERNEL_LANE_PER_PAIR,
                "agent": FLAG_KERNEL_LANE_PER_AGENT}


class Cm3Error(RuntimeError):
    pass


c_void_p, c_int32, c_int64, c_uint32, c_uint64 = (ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64,
                                                    ctypes.c_uint32, ctypes.c_uint64)
c_double, c_size_t = ctypes.c_double, ctypes.c_size_t


class ParticleDesc(ctypes.Structure):
    _fields_ = [("n_envs", c_int32), ("n_agents", c_int32), ("max_steps", c_int32), ("flags", c_uint32),
                ("env_offset", c_int32), ("env_count", c_int32),
                ("env_id_base", c_int64), ("seed", c_uint64), ("prob_random", c_double),
                ("initial_std", c_double),
                ("agents_x", c_double * MAX_AGENTS), ("agents_y", c_double * MAX_AGENTS),
                ("landmarks_x", c_double * MAX_AGENTS), ("landmarks_y", c_double * MAX_AGENTS)]


class ParticleBufs(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in (
        "state_in", "state_out", "goals_in", "goals_out", "meta_in", "meta_out", "episode", "actions",
        "obs_others", "reward_n", "reward", "done", "term_state", "term_obs_others", "collisions_tick")]


class ParticleTraj(ctypes.Structure):
    _fields_ = [("state", c_void_p), ("state_stride", c_size_t),
                ("goals", c_void_p), ("goals_stride", c_size_t),
                ("obs_others", c_void_p), ("obs_others_stride", c_size_t),
                ("actions", c_void_p), ("actions_stride", c_size_t),
                ("reward_n", c_void_p), ("reward_n_stride", c_size_t),
                ("reward", c_void_p), ("reward_stride", c_size_t),
                ("done", c_void_p), ("done_stride", c_size_t),
                ("meta", c_void_p), ("episode", c_void_p),
                ("term_state", c_void_p), ("term_state_stride", c_size_t),
                ("term_obs_others", c_void_p), ("term_obs_others_stride", c_size_t),
                ("collisions", c_void_p), ("collisions_stride", c_size_t),
                ("state_live", c_void_p), ("goals_live", c_void_p)]


class TransitionCols(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("state", "obs_others", "actions", "reward", "reward_n", "next_state", "next_obs_others",
                                        "done", "goals")] + [("ring_start", c_int64), ("ring_size", c_int64)]


class RowCols(ctypes.Structure):
    _fields_ = [("n_cols", c_int32), ("reserved", c_int32), ("dst", c_void_p * 16), ("src", c_void_p * 16),
                ("row_bytes", ctypes.c_uint32 * 16)]


class TileCol(ctypes.Structure):
    _fields_ = [("dst", c_void_p), ("src", c_void_p), ("n_rows", c_int64), ("elems_per_row", c_uint32), ("kind", c_uint32),
                ("elem_bytes", c_uint32), ("others_n", c_uint32), ("others_divq", c_uint32), ("others_divn", c_uint32),
                ("div", c_uint32 * 2), ("mod", c_uint32 * 2), ("mul", c_uint32 * 2)]


TILE_COPY, TILE_F32_TO_F64, TILE_ONEHOT_I64, TILE_ONEHOT_F64, TILE_EYE_F64, TILE_NOT_I64 = range(6)


class CopyShift(ctypes.Structure):
    _fields_ = [("n", c_int32), ("_pad", c_int32), ("first_dst", c_void_p * 4), ("mid", c_void_p * 4),
                ("last_src", c_void_p * 4), ("bytes", c_size_t * 4)]


class CheckersDesc(ctypes.Structure):
    _fields_ = [("n_envs", c_int32), ("n_agents", c_int32), ("n_rows", c_int32), ("n_columns", c_int32),
                ("n_obs", c_int32), ("max_steps", c_int32), ("flags", c_uint32), ("grid_stride", c_int32),
                ("obs_self_t_stride", c_int32), ("_pad", c_int32),
                ("env_id_base", c_int64), ("seed", c_uint64),
                ("agents_r", c_int32 * MAX_AGENTS), ("agents_c", c_int32 * MAX_AGENTS)]


class CheckersBufs(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in (
        "mask", "agents", "steps", "episode", "goals", "actions", "grid", "vec", "obs_others",
        "obs_self_t", "obs_self_v", "local_rewards", "reward", "done",
        "term_grid", "term_vec", "term_obs_others", "term_obs_self_t", "term_obs_self_v", "goals_next", "action_block")]


class CheckersTraj(ctypes.Structure):
    _fields_ = ([(n, c_void_p) for n in ("mask", "agents", "steps", "episode", "goals")] +
                [("actions", c_void_p), ("actions_stride", c_size_t), ("grid", c_void_p), ("grid_slot_stride", c_size_t),
                 ("vec", c_void_p), ("vec_stride", c_size_t), ("obs_others", c_void_p), ("obs_others_stride", c_size_t),
                 ("obs_self_t", c_void_p), ("obs_self_t_slot_stride", c_size_t),
                 ("obs_self_v", c_void_p), ("obs_self_v_stride", c_size_t),
                 ("local_rewards", c_void_p), ("local_rewards_stride", c_size_t),
                 ("reward", c_void_p), ("reward_stride", c_size_t), ("done", c_void_p), ("done_stride", c_size_t),
                 ("term_grid", c_void_p), ("term_grid_slot_stride", c_size_t),
                 ("term_vec", c_void_p), ("term_vec_stride", c_size_t),
                 ("term_obs_others", c_void_p), ("term_obs_others_stride", c_size_t),
                 ("term_obs_self_t", c_void_p), ("term_obs_self_t_slot_stride", c_size_t),
                 ("term_obs_self_v", c_void_p), ("term_obs_self_v_stride", c_size_t),
                 ("goals_slots", c_void_p), ("goals_slots_stride", c_size_t), ("action_block", c_void_p)])


class ActorParticleDesc(ctypes.Structure):
    _fields_ = [("n_envs", c_int32), ("n_agents", c_int32), ("stage", c_int32), ("n_h1_self", c_int32),
                ("n_h1_others", c_int32), ("n_h2", c_int32), ("n_actions", c_int32), ("epsilon", ctypes.c_float),
                ("precision", c_int32), ("_pad", c_int32), ("env_id_base", c_int64), ("seed", c_uint64)]


class ActorParticleWeights(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("w_self", "b_self", "w_self_h2", "w_others", "b_others", "w_others_h2",
                                        "b_h2", "w_out", "b_out", "packed")]


class ActorParticleBufs(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("obs_others", "state", "goals", "meta", "episode", "actions", "probs",
                                        "epsilon_dev")]


class ActorCheckersDesc(ctypes.Structure):
    _fields_ = [("n_envs", c_int32), ("n_agents", c_int32), ("stage", c_int32), ("n_obs", c_int32),
                ("conv_f", c_int32), ("n_conv_linear", c_int32), ("n_h1", c_int32), ("n_h2", c_int32),
                ("n_actions", c_int32), ("epsilon", ctypes.c_float), ("precision", c_int32),
                ("obs_self_t_stride", c_int32), ("env_id_base", c_int64), ("seed", c_uint64)]


class ActorCheckersWeights(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("conv_w", "conv_b", "lin_w", "lin_b", "self_w", "self_b", "w_self_h2",
                                        "others_w", "others_b", "w_others_h2", "b_h2", "out_w", "out_b", "packed")]


class ActorCheckersBufs(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("obs_self_t", "obs_self_v", "obs_others", "goals", "actions_prev", "steps",
                                        "episode", "actions", "probs", "prev_done", "epsilon_dev")]


# every symbol include/cm3_amd.h declares: name -> (restype, argtypes)
P = ctypes.POINTER
SYMBOLS = {
    "cm3_abi_version": (ctypes.c_int, []),
    "cm3_source_id": (ctypes.c_char_p, []),
    "cm3_last_error": (ctypes.c_char_p, []),
    "cm3_last_kernel_variant": (ctypes.c_char_p, []),
    "cm3_device_count": (ctypes.c_int, []),
    "cm3_device_name": (ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, ctypes.c_int]),
    "cm3_particle_step_f32": (ctypes.c_int, [P(ParticleDesc), P(ParticleBufs), c_void_p]),
    "cm3_particle_step_f64": (ctypes.c_int, [P(ParticleDesc), P(ParticleBufs), c_void_p]),
    "cm3_particle_reset_f32": (ctypes.c_int, [P(ParticleDesc), P(ParticleBufs), c_void_p, c_void_p]),
    "cm3_particle_reset_f64": (ctypes.c_int, [P(ParticleDesc), P(ParticleBufs), c_void_p, c_void_p]),
    "cm3_particle_observe_f32": (ctypes.c_int, [P(ParticleDesc), P(ParticleBufs), c_void_p]),
    "cm3_particle_observe_f64": (ctypes.c_int, [P(ParticleDesc), P(ParticleBufs), c_void_p]),
    "cm3_particle_rollout_f32": (ctypes.c_int, [P(ParticleDesc), P(ParticleTraj), c_int32, c_void_p]),
    "cm3_particle_rollout_f64": (ctypes.c_int, [P(ParticleDesc), P(ParticleTraj), c_int32, c_void_p]),
    "cm3_checkers_step": (ctypes.c_int, [P(CheckersDesc), P(CheckersBufs), c_void_p]),
    "cm3_checkers_rollout": (ctypes.c_int, [P(CheckersDesc), P(CheckersTraj), c_int32, c_void_p]),
    "cm3_checkers_reset": (ctypes.c_int, [P(CheckersDesc), P(CheckersBufs), c_void_p, c_void_p]),
    "cm3_checkers_action_blocks": (ctypes.c_int, [P(CheckersDesc), c_void_p, c_void_p]),
    "cm3_actor_particle_packed_bytes": (c_size_t, [c_int32]),
    "cm3_actor_particle_pack": (ctypes.c_int, [P(ActorParticleDesc), P(ActorParticleWeights), c_void_p, c_void_p]),
    "cm3_actor_particle_f32": (ctypes.c_int, [P(ActorParticleDesc), P(ActorParticleWeights), P(ActorParticleBufs),
                                              c_void_p]),
    "cm3_policy_rollout_f32": (ctypes.c_int, [P(ParticleDesc), P(ParticleTraj), P(ActorParticleDesc),
                                              P(ActorParticleWeights), c_void_p, c_size_t, c_int32, c_void_p]),
    "cm3_policy_force_row_tiles": (ctypes.c_int, [c_int32]),
    "cm3_td_target_f64": (ctypes.c_int, [c_void_p, c_int32, c_void_p, c_void_p, ctypes.c_double, c_void_p, c_int64, c_void_p]),
    "cm3_transitions_gather_f32": (ctypes.c_int, [P(ParticleDesc), P(ParticleTraj), c_void_p, c_size_t, c_void_p, c_void_p, c_int64,
                                                  P(TransitionCols), c_void_p]),
    "cm3_rows_scatter": (ctypes.c_int, [P(RowCols), c_int64, c_void_p, c_int64, c_int64, c_void_p]),
    "cm3_rows_gather": (ctypes.c_int, [P(RowCols), c_int64, c_void_p, c_void_p]),
    "cm3_rows_tile": (ctypes.c_int, [P(TileCol), c_int32, c_void_p]),
    "cm3_actor_checkers_packed_bytes": (c_size_t, []),
    "cm3_actor_checkers_pack": (ctypes.c_int, [P(ActorCheckersDesc), P(ActorCheckersWeights), c_void_p, c_void_p]),
    "cm3_policy_rollout_checkers": (ctypes.c_int, [P(CheckersDesc), P(CheckersTraj), P(ActorCheckersDesc), P(ActorCheckersWeights),
                                                   c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, P(CheckersBufs), c_int32, c_void_p]),
    "cm3_actor_checkers_f32": (ctypes.c_int, [P(ActorCheckersDesc), P(ActorCheckersWeights), P(ActorCheckersBufs),
                                              c_void_p]),
    "cm3_returns_scratch_bytes": (c_size_t, []),
    "cm3_returns_moments_f32": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_int32, c_int32, c_int32, c_double, c_void_p]),
    "cm3_returns_moments_f64": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_int32, c_int32, c_int32, c_double, c_void_p]),
    "cm3_normalize_f32": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_size_t, c_int32, c_double,
                                         c_int32, c_void_p]),
    "cm3_normalize_f64": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_size_t, c_int32, c_double,
                                         c_int32, c_void_p]),
    "cm3_returns_normalize_f32": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                 c_int32, c_int32, c_int32, c_double, c_double, c_int32, c_void_p, c_void_p]),
    "cm3_returns_normalize_f64": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                 c_int32, c_int32, c_int32, c_double, c_double, c_int32, c_void_p, c_void_p]),
    "cm3_returns_segments_scratch_bytes": (c_size_t, [c_int32]),
    "cm3_returns_normalize_segments_f32": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                          c_int32, c_int32, c_int32, c_int32, c_double, c_double, c_int32, c_void_p,
                                                          c_void_p]),
    "cm3_returns_normalize_segments_f64": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                          c_int32, c_int32, c_int32, c_int32, c_double, c_double, c_int32, c_void_p,
                                                          c_void_p]),
    "cm3_normalize_segments_f32": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_size_t, c_int32,
                                                  c_double, c_int32, c_void_p]),
    "cm3_normalize_segments_f64": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_size_t, c_int32,
                                                  c_double, c_int32, c_void_p]),
    "cm3_copy_list": (ctypes.c_int, [c_int32, P(c_void_p), P(c_void_p), P(c_size_t), c_void_p]),
    "cm3_hbm_read_bench": (ctypes.c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),
    "cm3_hbm_bench_sink_words": (ctypes.c_int, []),
    "cm3_hbm_read_bench_cfg": (ctypes.c_int, [c_void_p, c_size_t, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "cm3_hbm_copy_bench": (ctypes.c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "cm3_hbm_copy_bench_cfg": (ctypes.c_int, [c_void_p, c_void_p, c_size_t, c_int32, c_int32, c_int32, c_void_p]),
    "cm3_traffic_floor_bench": (ctypes.c_int, [c_void_p, c_size_t, c_void_p, c_size_t, c_int32, c_int32, c_void_p]),
    "cm3_graph_begin": (ctypes.c_int, [c_void_p]),
    "cm3_graph_end": (ctypes.c_int, [c_void_p, P(c_void_p)]),
    "cm3_graph_launch": (ctypes.c_int, [c_void_p, c_void_p]),
    "cm3_graph_destroy": (ctypes.c_int, [c_void_p]),
    "cm3_event_create": (ctypes.c_int, [P(c_void_p)]),
    "cm3_event_record": (ctypes.c_int, [c_void_p, c_void_p]),
    "cm3_event_synchronize": (ctypes.c_int, [c_void_p]),
    "cm3_event_elapsed_ms": (ctypes.c_int, [c_void_p, c_void_p, P(ctypes.c_float)]),
    "cm3_event_destroy": (ctypes.c_int, [c_void_p]),
    "cm3_stream_synchronize": (ctypes.c_int, [c_void_p]),
}

_lib = None


def lib():
    """The loaded library (loads on first use).  Raises Cm3Error if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch bundles its own libamdhip64.so.7; load it FIRST so that this library binds to the same HIP
    # runtime that owns the tensors' memory and streams (two runtimes in one process see no device).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise Cm3Error("HIP extension not built: %s is missing.  Run `python -c 'import __graft_entry__ as g; "
                       "g.build()'` (or cm3_amd/csrc/build.sh).  There is no CPU fallback." % LIB_PATH)
    try:
        handle = ctypes.CDLL(LIB_PATH)
    except OSError as exc:
        raise Cm3Error("failed to load %s: %s" % (LIB_PATH, exc))
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(handle, name)
        except AttributeError:
            raise Cm3Error("libcm3_hip.so does not export %s (stale build?)" % name)
        fn.restype = res
        fn.argtypes = args
    if handle.cm3_abi_version() != ABI_VERSION:
        raise Cm3Error("ABI version mismatch: library %d, binding %d" % (handle.cm3_abi_version(), ABI_VERSION))
    # a library built from other sources than the ones next to this file is refused (a test run against a stale build proves
    # nothing); CM3_AMD_LIB = an explicitly chosen build (same-box A/B comparisons, the two-stamp build) is taken as it is
    if "CM3_AMD_LIB" not in os.environ and os.environ.get("CM3_AMD_ALLOW_STALE") != "1":
        built, have = handle.cm3_source_id().decode(), source_id()
        if have is not None and built != have:
            raise Cm3Error("%s was built from other sources (library %s, sources %s): rebuild with cm3_amd/csrc/build.sh "
                           "or `python -c 'import __graft_entry__ as g; g.build()'`" % (LIB_PATH, built, have))
    _lib = handle
    return _lib


def source_id():
    """What csrc/build.sh bakes into the library as cm3_source_id(): the first 16 hex digits of the SHA-256 over csrc/*.hip and
    csrc/*.h (byte order of the names) followed by include/cm3_amd.h.  None when the sources are not there (a wheel without them)."""
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    csrc = os.path.join(here, "csrc")
    header = os.path.join(os.path.dirname(here), "include", "cm3_amd.h")
    if not os.path.isdir(csrc) or not os.path.exists(header):
        return None
    names = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h")))
    h = hashlib.sha256()
    for f in names + [header]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def last_kernel_variant():
    """The kernel instantiation the most recent env call of this thread launched (cm3_last_kernel_variant): a debug / test query."""
    return lib().cm3_last_kernel_variant().decode()


def check(rc):
    if rc != 0:
        msg = lib().cm3_last_error()
        raise Cm3Error("libcm3_hip error %d: %s" % (rc, msg.decode("utf-8", "replace") if msg else "?"))


def ptr(t):
    """Device pointer of a torch tensor (0 for None)."""
    return 0 if t is None else t.data_ptr()


def copy_list(pairs, stream):
    """ONE launch copying every (dst tensor, src tensor) pair (cm3_copy_list; 16-byte aligned contiguous tensors)."""
    n = len(pairs)
    dst = (c_void_p * n)(*[d.data_ptr() for d, _ in pairs])
    src = (c_void_p * n)(*[s.data_ptr() for _, s in pairs])
    nb = (c_size_t * n)(*[d.numel() * d.element_size() for d, _ in pairs])
    check(lib().cm3_copy_list(n, dst, src, nb, stream))


def row_cols(pairs):
    """cm3_row_cols for (dst tensor, src tensor) pairs: rows = leading dim, row_bytes from the trailing dims (contiguous tensors)."""
    if not 1 <= len(pairs) <= 16:
        raise Cm3Error("1..16 columns per launch")
    rc = RowCols()
    rc.n_cols = len(pairs)
    for k, (d, s_) in enumerate(pairs):
        if not (d.is_contiguous() and s_.is_contiguous()) or d.dtype != s_.dtype or tuple(d.shape[1:]) != tuple(s_.shape[1:]):
            raise Cm3Error("row columns must be contiguous and agree in dtype / row shape (column %d)" % k)
        rc.dst[k], rc.src[k] = d.data_ptr(), s_.data_ptr()
        rb = d.element_size()
        for n in d.shape[1:]:
            rb *= int(n)
        rc.row_bytes[k] = rb
    return rc


def rows_scatter(pairs, n_rows, stream, dst_row=None, ring_start=0, ring_size=0):
    """ONE launch: dst[row(b)] = src[b] for every (dst, src) column pair (cm3_rows_scatter)."""
    for k in range(0, len(pairs), 16):
        rc = row_cols(pairs[k:k + 16])
        check(lib().cm3_rows_scatter(ctypes.byref(rc), int(n_rows), ptr(dst_row), int(ring_start), int(ring_size), stream))


def rows_gather(pairs, n_rows, src_row, stream):
    """ONE launch: dst[b] = src[src_row[b]] for every (dst, src) column pair (cm3_rows_gather)."""
    for k in range(0, len(pairs), 16):
        rc = row_cols(pairs[k:k + 16])
        check(lib().cm3_rows_gather(ctypes.byref(rc), int(n_rows), ptr(src_row), stream))


def current_stream_handle(device):
    import torch
    return torch.cuda.current_stream(device).cuda_stream


def capture_graph(device, enqueue):
    """Captures whatever ``enqueue(stream_handle)`` launches into a hipGraph and returns the executable
    graph handle.  Capture happens on a private side stream (the legacy default stream cannot be captured);
    the graph can then be launched on any stream with cm3_graph_launch."""
    import torch
    side = torch.cuda.Stream(device=device)
    handle = ctypes.c_void_p()
    check(lib().cm3_graph_begin(side.cuda_stream))
    try:
        enqueue(side.cuda_stream)
    finally:
        rc = lib().cm3_graph_end(side.cuda_stream, ctypes.byref(handle))
    check(rc)
    return handle


def require_gpu(device):
    """Fail loudly when asked to compute without a GPU: there is no CPU path in the product."""
    import torch
    dev = torch.device(device)
    if dev.type != "cuda":
        raise Cm3Error("cm3_amd computes only on an AMD GPU (device=%r given); there is no CPU fallback" % (device,))
    if not torch.cuda.is_available():
        raise Cm3Error("no HIP device visible to PyTorch; cm3_amd has no CPU fallback")
    lib()
    return dev
