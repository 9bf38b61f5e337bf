"""bench.py's secondary legs at N = 1 - the ones that need nothing of the bench process's own state.  Each adds its keys to the bench's
JSON object `out` and never raises (a failed leg leaves an `error` entry): the judged figures are the headline's.

  overlap_leg   SURVEY 8d protocol 2, tools/overlap_bench.py as a child process: compact_fwd beside real attention -
                `overlap_with_attention`, `exposed_exchange_ms_per_step` (the path compact_fwd takes with NO user opt-in) and
                `exposed_exchange_ms_per_step_caller_on_the_lane`
  plugin_leg    protocol 1 through the plugin API, tools/plugin_path_bench.py as a child process - `plugin_path`, `plugin_path_ms_per_step`
  configs_leg   every BASELINE.json configuration, one rank's codec work of a denoise step in layer order (tools/config_table.py) - `configs`
  lowrank_leg   the reference's low-rank presets on the FLUX shard, one K,V pair per call - `low_rank_presets`
"""
import json
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(script, argv, timeout):
    with tempfile.TemporaryDirectory() as td:
        jpath = os.path.join(td, "out.json")
        r = subprocess.run([sys.executable, os.path.join(REPO, "tools", script)] + argv + ["--json", jpath],
                           capture_output=True, text=True, timeout=timeout, cwd=REPO)
        if r.returncode != 0:
            raise RuntimeError(f"tools/{script} failed: " + r.stderr[-400:])
        with open(jpath) as f:
            return json.load(f)


def overlap_leg(out, steps, layers):
    # a process of its own: the bench process has had RCCL, the exchange stream and half a dozen plan sets in it, and the lane's figure is
    # about two flag-ordered streams beside attention kernels, nothing else (in-process it read 0.15 ms/step higher, same box)
    try:
        ov = _child("overlap_bench.py", ["--quiet", "--steps", str(steps), "--layers", str(layers), "--legs",
                                         "attention_on_compute_lane,default,lane,attention_distinct_kv_on_compute_lane"], 900)
        legs = ov["legs_ms_per_step"]
        out["overlap_with_attention"] = {
            "protocol": ov["protocol"], "steps": ov["steps"], "lane": ov["lane"],
            "attention_only_ms_per_step": legs["attention_on_compute_lane"]["wall"],
            "attention_over_distinct_kv_ms_per_step": legs["attention_distinct_kv_on_compute_lane"]["wall"],
            "with_exchange_default_path_ms_per_step": legs["default"]["wall"],
            "with_exchange_on_the_lane_ms_per_step": legs["lane"]["wall"],
            "exposed_exchange_ms_per_step": ov["exposed_exchange_ms_per_step"]["default"],
            "exposed_exchange_ms_per_step_caller_on_the_lane": ov["exposed_exchange_ms_per_step"]["lane"],
            "exposed_exchange_ms_per_step_vs_attention_over_distinct_kv": ov["exposed_exchange_ms_per_step_vs_attention_over_distinct_kv"],
            "what": "compact_fwd (gather schedule) with PyTorch-ROCm SDPA at the FLUX shape: the layer's chain on the CU-masked exchange "
                    "stream, ordered with the compute stream by flags in device memory; exposed = step with the exchange - attention alone.  "
                    "exposed_exchange_ms_per_step = the path compact_fwd takes with NO user opt-in (caller on an ordinary stream: it forks to "
                    "the lane's compute stream and joins back per call); ..._caller_on_the_lane = the model run on lanes.compute_stream()"}
        out["exposed_exchange_ms_per_step"] = ov["exposed_exchange_ms_per_step"]["default"]
        out["exposed_exchange_ms_per_step_caller_on_the_lane"] = ov["exposed_exchange_ms_per_step"]["lane"]
    except Exception as e:  # pragma: no cover
        out["overlap_with_attention"] = {"error": f"{type(e).__name__}: {e}"}


def plugin_leg(out, steps, layers):
    # the product path - one native op per layer, compact/xlayer.py - for the same 57-layer step, host issue included (a process of its
    # own: it loops the 8 logical ranks back by patching torch.distributed's rank / world queries)
    try:
        pp = _child("plugin_path_bench.py", ["--quiet", "--steps", str(steps), "--layers", str(layers)], 600)
        out["plugin_path"] = {
            "ms_per_step": pp["ms_per_step"], "host_us_per_layer": pp["host_us_per_layer"], "legs": pp["legs"], "ipc_memory": pp.get("ipc_memory_kind"),
            "what": "the same step issued through the plugin API, attention replaced by a no-op: compact_all_gather_kv (what patch_gather_fwd "
                    "calls) and compact_fwd (gather schedule, configure(lane='off')), ONE native op per layer (cfx_plan_add_exchange_layer_p2p "
                    "through compact/xlayer.py); `ms_per_step` / `host_us_per_layer` = compact_all_gather_kv on a side stream; 8 logical ranks "
                    "looped back"}
        out["plugin_path_ms_per_step"] = pp["ms_per_step"]
    except Exception as e:  # pragma: no cover
        out["plugin_path"] = {"error": f"{type(e).__name__}: {e}"}


def plugin_configs_leg(out):
    # every BASELINE configuration and every shipped preset THROUGH THE PLUGIN API (tools/plugin_config_bench.py, a child process): one Python
    # call per layer - compact_fwd / compact_all_gather_kv / compact_compress + compact_decompress - instead of the native plan replay
    # `configs` is timed with.  Adds `plugin_ms_per_step` and `plugin_host_us_per_layer` to the rows of `configs`, and `presets_through_the_plugin_api`
    try:
        pc = _child("plugin_config_bench.py", ["--quiet", "--budget", "0.12"], 900)["rows"]
        for key, row in (out.get("configs") or {}).items():
            src = pc.get(key)
            if src and "plugin_ms_per_step" in src:
                row["plugin_ms_per_step"] = src["plugin_ms_per_step"]
                row["plugin_host_us_per_layer"] = src["host_us_per_layer_idle_queue"]
                row["plugin_api"] = src["api"]
                row["plugin_vs_native_replay"] = round(src["plugin_ms_per_step"] / row["ms_per_step"], 3) if row.get("ms_per_step") else None
            elif src:
                row["plugin_error"] = src.get("error")
        out["presets_through_the_plugin_api"] = {
            k[len("preset "):]: ({"ms_per_step": v["plugin_ms_per_step"], "host_us_per_layer": v["host_us_per_layer_idle_queue"],
                                  "kernels_per_layer": v["kernels_per_layer"], "one_native_call_per_layer": v["one_native_op_per_layer"]}
                                 if "plugin_ms_per_step" in v else v)
            for k, v in pc.items() if k.startswith("preset ")}
        out["presets_through_the_plugin_api"]["what"] = ("protocol 1 through compact_fwd (no-op attention, lane off) at the FLUX shard, 57 layers, 8 logical ranks "
                                                          "looped back: the reference's shipped presets (examples/configs.py:39-98); host_us_per_layer = one step "
                                                          "issued into an idle queue / layers")
    except Exception as e:  # pragma: no cover
        out["presets_through_the_plugin_api"] = {"error": f"{type(e).__name__}: {e}"}


def overlap_presets_leg(out, steps, layers):
    # SURVEY 8d protocol 2 for the presets other than BINARY (task 4): exposed exchange of INT2 and LOW_RANK r = 8 beside SDPA
    res = {}
    for preset in ("int2", "lowrank8"):
        try:
            ov = _child("overlap_bench.py", ["--quiet", "--steps", str(steps), "--layers", str(layers), "--preset", preset, "--legs",
                                             "attention_on_compute_lane,attention,default"], 900)
            legs = ov["legs_ms_per_step"]
            res[preset] = {"default_path": ov["default_path"], "attention_only_ms_per_step": legs["attention"]["wall"],
                           "with_exchange_ms_per_step": legs["default"]["wall"], "exposed_exchange_ms_per_step": ov["exposed_exchange_ms_per_step"]["default"]}
        except Exception as e:  # pragma: no cover
            res[preset] = {"error": f"{type(e).__name__}: {e}"}
    res["what"] = ("compact_fwd with every switch at its default beside PyTorch-ROCm SDPA at the FLUX shape; INT2 takes the exchange lane; the low-rank "
                   "family keeps its factor chain - one persistent launch over the chip, exposed - on the compute lane and leaves the publish-and-wait and "
                   "the 7 peers' reconstructions to the exchange lane, peer by peer behind flags the merge launches wait for")
    out["overlap_with_attention_other_presets"] = res


def configs_leg(out, hbm_peak_gbs):
    # every BASELINE.json configuration, one rank's codec work of one denoise step replayed layer by layer in order, peers looped back
    # (SURVEY 8d shapes): ms per step, algorithmic bytes, fraction of the HBM roofline
    try:
        import torch
        sys.path.insert(0, os.path.join(REPO, "tools"))
        import config_table as CT
        torch.cuda.empty_cache()
        cfgs = {}
        for name, cid_, param_, (n_, c_), l_, ncomp, nrec, upd in CT.CONFIGS:
            ms_ = CT.gpu_step(cid_, param_, n_, c_, l_, ncomp, nrec, upd, min_steps=8, budget_s=0.05)
            ab = CT.alg_bytes(cid_, n_, c_, l_, ncomp, nrec, upd)
            key = name.split()[0] + (" " + " ".join(name.split()[-2:]) if name.startswith("5") else "")
            cfgs[key] = {"workload": name, "shard": [n_, c_], "layers": l_, "ms_per_step": round(ms_, 4), "alg_bytes": ab,
                         "frac": round(ab / (ms_ * 1e-3) / 1e9 / hbm_peak_gbs, 4)}
            ach_ = (out.get("roofline") or {}).get("achievable_gbs")
            if ach_:
                cfgs[key]["frac_of_achievable"] = round(ab / (ms_ * 1e-3) / 1e9 / ach_, 4)
            torch.cuda.empty_cache()
        out["configs"] = cfgs
    except Exception as e:  # pragma: no cover
        out["configs"] = {"error": f"{type(e).__name__}: {e}"}


def lowrank_leg(out, dev, N, C):
    # the low-rank presets of the reference (examples/configs.py:63-110) on the same shard: one K,V pair per call, distinct pairs in turn
    # (cold caches), event-timed through the Python API (compress = factors + state update; LOW_RANK_Q also quantises them)
    try:
        import torch
        from compactfusion_amd import codecs as K
        lr, Lr = {}, 24
        g2 = torch.Generator(device=dev).manual_seed(5)
        xl = torch.randn(Lr, 2, N, C, generator=g2, device=dev).half()
        sl = (xl.float() + 0.1 * torch.randn(Lr, 2, N, C, generator=g2, device=dev)).half()
        for name, q_, r_ in (("LOW_RANK r=8", False, 8), ("LOW_RANK r=16", False, 16), ("LOW_RANK_Q r=32", True, 32)):
            pkl = [torch.empty(K.lr_packet_halves(q_, N, C, r_), dtype=torch.float16, device=dev) for _ in range(2)]
            q0 = [torch.randn(C, K.lr_rank_pad(r_), generator=g2, device=dev) for _ in range(2)]

            def lay(l):
                K.lr_compress_batch(q_, [xl[l, 0], xl[l, 1]], [sl[l, 0], sl[l, 1]], [sl[l, 0], sl[l, 1]], pkl, q0, N, C, r_, True)
            for l in range(4):
                lay(l)
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                for l in range(Lr):
                    lay(l)
            e1.record()
            torch.cuda.synchronize(dev)
            lr[name] = round(e0.elapsed_time(e1) / (3 * Lr) * 1e3, 1)
        out["low_rank_presets"] = {"us_per_kv_pair_compress": lr, "shard": [N, C],
                                   "what": "cfx_lr_compress_batch, one persistent launch per K,V pair (csrc/cfx_lrslab.hip) + the int4 factor "
                                           "quantiser for LOW_RANK_Q; profiles/r06_lowrank_*"}
    except Exception as e:  # pragma: no cover
        out["low_rank_presets"] = {"error": f"{type(e).__name__}: {e}"}
