"""One place for the host-side switches of the package: `compactfusion_amd.configure(...)`.

Every setting has a documented default; an environment variable (the round 1-4 switches, kept as a thin default layer so that a
launcher script can set them without touching code) overrides the default, and an explicit `configure()` call overrides both.
Settings are read when they are used, so `configure()` may be called at any time before the layers they affect are (re)built
(`compact_init` / `compact_reset` rebuild them).

    import compactfusion_amd
    compactfusion_amd.configure(exchange="rccl", lane="sticky", lane_exchange_cus=48)

setting                env var                       values / meaning
---------------------  ----------------------------  ---------------------------------------------------------------------------------
exchange               CFX_EXCHANGE                  auto | p2p | rccl | torch - transport of the one-op layer exchange (compact/xlayer.py)
ring_schedule          CFX_RING_SCHEDULE             auto | gather | relay - exchange schedule of compact_fwd (auto: gather on a GPU)
ring_exchange          CFX_RING_EXCHANGE             auto | native | torch - who issues the collective of the multi-launch schedules
ring_exchange_stream   CFX_RING_EXCHANGE_STREAM      auto | xlayer | lane | chain | side | main - where a layer's exchange chain runs
ring_p2p               CFX_RING_P2P                  (unset) | 0 | 1 - 1: the multi-launch chain reads packets in place through IPC mappings too;
                                                     0: no peer-to-peer transport at all (the layer op starts at rccl)
ring_exchange_priority CFX_RING_EXCHANGE_PRIORITY    priority of the (unmasked) exchange stream of the event-ordered schedules (-1)
lane                   CFX_LANE                      auto | sticky | off - how compact_fwd gets onto the exchange lane when the caller is not
                                                     on it already: auto = for the duration of the call (forked from and joined to the
                                                     caller's stream with flag kernels: the model's other kernels keep the caller's stream);
                                                     sticky = the caller's current stream BECOMES the lane's compute stream at the first
                                                     call and stays it (no per-layer hand-over; the rest of the model runs on 224 CUs);
                                                     off = never (the exchange runs as one op on the caller's stream, nothing overlaps)
lane_exchange_cus      CFX_LANE_EXCHANGE_CUS         CUs of the exchange lane (32); the compute lane gets the rest
lowrank_lane           CFX_LOWRANK_LANE              on | off - on: a LOW_RANK / LOW_RANK_Q layer keeps its factor chain (one persistent launch that
                                                     wants the chip) on the compute lane and puts the peers' reconstructions on the exchange lane
                                                     beside the attention blocks; off: the whole layer op on the caller's stream
hw_queues              GPU_MAX_HW_QUEUES             hardware queues HIP may give its streams - HIP reads it ONCE when it initialises:
                                                     configure(hw_queues=8) must run before the first CUDA call of the process
"""
from __future__ import annotations

import os
from typing import Any, Dict

_SETTINGS = {
    # name: (env var, default, allowed values or None)
    "exchange": ("CFX_EXCHANGE", "auto", ("auto", "p2p", "rccl", "torch")),
    "ring_schedule": ("CFX_RING_SCHEDULE", "auto", ("auto", "gather", "relay")),
    "ring_exchange": ("CFX_RING_EXCHANGE", "auto", ("auto", "native", "torch")),
    "ring_exchange_stream": ("CFX_RING_EXCHANGE_STREAM", "auto", ("auto", "xlayer", "lane", "chain", "side", "main")),
    "ring_p2p": ("CFX_RING_P2P", "", None),
    "ring_exchange_priority": ("CFX_RING_EXCHANGE_PRIORITY", "-1", None),
    "lane": ("CFX_LANE", "auto", ("auto", "sticky", "off")),
    "lane_exchange_cus": ("CFX_LANE_EXCHANGE_CUS", "32", None),
    "lowrank_lane": ("CFX_LOWRANK_LANE", "on", ("on", "off")),
}
_explicit: Dict[str, str] = {}


def configure(**kw: Any) -> Dict[str, str]:
    """Set host-side switches (see the module docstring); returns the effective settings.  `hw_queues=n` exports GPU_MAX_HW_QUEUES
    for the HIP runtime - it must be called before the process's first CUDA call and raises otherwise."""
    for name, value in kw.items():
        if name == "hw_queues":
            _set_hw_queues(int(value))
            continue
        if name not in _SETTINGS:
            raise TypeError(f"configure() has no setting {name!r}; settings: {', '.join(sorted(_SETTINGS))}, hw_queues")
        allowed = _SETTINGS[name][2]
        value = str(int(value)) if isinstance(value, bool) else str(value)
        if allowed is not None and value not in allowed:
            raise ValueError(f"{name} must be one of {' | '.join(allowed)} (got {value!r})")
        _explicit[name] = value
    return {n: get(n) for n in _SETTINGS}


def get(name: str) -> str:
    env, default, allowed = _SETTINGS[name]
    if name in _explicit:
        return _explicit[name]
    v = os.environ.get(env, default)
    if allowed is not None and v not in allowed:
        raise ValueError(f"{env} must be one of {' | '.join(allowed)} (got {v!r})")
    return v


def reset() -> None:
    """Forget every explicit setting (tests)."""
    _explicit.clear()


def _hip_is_up() -> bool:
    import sys
    torch = sys.modules.get("torch")
    try:
        return torch is not None and torch.cuda.is_initialized()
    except Exception:  # noqa: BLE001
        return True


def _set_hw_queues(n: int) -> None:
    if n < 1:
        raise ValueError("hw_queues must be positive")
    if _hip_is_up() and os.environ.get("GPU_MAX_HW_QUEUES") != str(n):
        raise RuntimeError("configure(hw_queues=...) must run before the first CUDA call of the process: HIP reads GPU_MAX_HW_QUEUES once")
    os.environ["GPU_MAX_HW_QUEUES"] = str(n)
