"""FLUX.1-dev with the residual-compressed K,V exchange running on libcfx - the reference's examples/flux_example.py:92-127 flow with the
`xfuser.compact` package replaced by `compactfusion_amd` (compat.install_xfuser_alias), plus the quality band BASELINE.json asks for: PSNR
(and LPIPS when the `lpips` package is importable) of every compressed image against the uncompressed ring-attention image of the same seed.

NOT runnable in the build / test image of this repository: it needs a CompactFusion (xDiT) checkout on PYTHONPATH, diffusers, the FLUX.1-dev
weights and one process per GPU, none of which exist there - the codec-level quality traces the repository does pin are
tests/test_gpu_quality.py (G12) and tests/test_gpu_stack.py (G13).  On a machine that has them:

    torchrun --nproc-per-node 8 examples/flux_compact_example.py --model black-forest-labs/FLUX.1-dev --ring-degree 8 \\
        --preset binary --prompt "a photo of a cat" --out out/

Presets follow the reference's examples/configs.py:39-98 (1 WARMUP step, residual 1 + error feedback).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compactfusion_amd  # noqa: E402  (exports GPU_MAX_HW_QUEUES before HIP starts)
from compactfusion_amd.compat import install_xfuser_alias  # noqa: E402

install_xfuser_alias()          # before anything imports xfuser.compact

import torch  # noqa: E402


def preset(name):
    """the reference's named configurations (examples/configs.py there; compactfusion_amd/compact/presets.py here)"""
    from xfuser.compact.presets import get_config                                      # = compactfusion_amd.compact.presets
    return get_config("Flux", "ring" if name == "off" else name)


def psnr(a, b):
    mse = torch.mean((a.float() - b.float()) ** 2)
    return float(10 * torch.log10(1.0 / mse)) if mse > 0 else float("inf")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="black-forest-labs/FLUX.1-dev")
    ap.add_argument("--prompt", default="a photo of an astronaut riding a horse on the moon")
    ap.add_argument("--ring-degree", type=int, default=int(os.environ.get("WORLD_SIZE", "1")))
    ap.add_argument("--preset", default="binary", choices=["binary", "int2", "lowrank8", "lowrank12", "lowrankq32"])
    ap.add_argument("--steps", type=int, default=28)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--out", default="out")
    ap.add_argument("--lane", default="auto", choices=["auto", "sticky", "off"])
    args = ap.parse_args()
    compactfusion_amd.configure(lane=args.lane)

    try:        # everything below is the reference's own driver flow (examples/flux_example.py), unchanged
        from xfuser import xFuserArgs, xFuserFluxPipeline
        from xfuser.config import FlexibleArgumentParser
        from xfuser.core.distributed import get_world_group
    except ImportError as e:
        raise SystemExit(f"this example needs a CompactFusion / xDiT checkout and diffusers on PYTHONPATH ({e}); see the module docstring") from e
    from xfuser.compact.main import compact_init, compact_reset, compact_hello
    from xfuser.collector.collector import Collector, init as collector_init

    engine_args = xFuserArgs.from_cli_args(FlexibleArgumentParser().parse_args(
        ["--model", args.model, "--ring_degree", str(args.ring_degree), "--num_inference_steps", str(args.steps), "--prompt", args.prompt,
         "--height", "1024", "--width", "1024", "--seed", str(args.seed)]))
    engine_config, input_config = engine_args.create_config()
    local_rank = get_world_group().local_rank
    os.makedirs(args.out, exist_ok=True)
    collector_init(Collector(os.path.join(args.out, "collector"), enabled=False))
    images = {}
    for name in ("off", args.preset):
        compact_init(preset(name))                       # must precede the model build: attn_layer.py:59-64 binds compact_fwd once
        pipe = xFuserFluxPipeline.from_pretrained(pretrained_model_name_or_path=engine_config.model_config.model, engine_config=engine_config,
                                                  torch_dtype=torch.bfloat16).to(f"cuda:{local_rank}")
        pipe.prepare_run(input_config)
        compact_hello()
        compact_reset()
        out = pipe(height=1024, width=1024, prompt=args.prompt, num_inference_steps=args.steps, output_type="pt",
                   generator=torch.Generator(device="cuda").manual_seed(args.seed))
        images[name] = out.images[0].float().cpu()
        del pipe
        torch.cuda.empty_cache()
    if get_world_group().rank == 0:
        ref, img = images["off"], images[args.preset]
        line = f"{args.preset}: PSNR vs uncompressed ring attention {psnr(img, ref):.2f} dB"
        try:
            import lpips
            d = lpips.LPIPS(net="alex")(img.unsqueeze(0) * 2 - 1, ref.unsqueeze(0) * 2 - 1)
            line += f", LPIPS {float(d):.4f}"
        except ImportError:
            line += " (lpips not installed: no LPIPS)"
        print(line)
        torch.save(images, os.path.join(args.out, "images.pt"))


if __name__ == "__main__":
    main()
