#!/usr/bin/env python3
"""Benchmark of the VL-CABS zero-shot classification hot path (BASELINE.json metric):
images/sec (+ similarity-maps/sec) on synthetic 1024x1024 CXR x 14 prompts, bf16, B=32 per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One step = one compute_logits-equivalent pass (vision encoder + VL-CABS head) over one batch of B images per
GPU, pixels already resident in HBM, prompt embeddings cached (encoded once, sharded over ranks + one RCCL
all_gather, before the timed region: their one-time cost is reported separately).  Prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from radzero_amd.config import RadZeroConfig, attention_flops_per_image_layer, flops_per_image  # noqa: E402
from radzero_amd.synthetic import synthetic_prompts  # noqa: E402
from radzero_amd.weights import make_state_dict  # noqa: E402

# Hardware peaks (MI355X_MICROARCH.md): 2.5 PFLOP/s dense MFMA for bf16 / f16 operands — EVERY roofline block below prints `peak` = 2500 and `frac` against it,
# the fp32 mode's too.  The fp32 mode computes every product on the 16-bit / fp8 matrix pipes over split operands, so one
# algorithmic product issues more than one MFMA unit (one unit = one v_mfma_f32_16x16x32_f16, 16 cycles):
#   three-plane form   a_hi b_hi + a_lo b_hi + a_hi b_lo as three f16 MFMAs                                                   -> 3 units
#   MX form            a_hi b_hi on the f16 pipe + both correction terms as ONE block-scaled e4m3 MFMA over K' = 128 (32 cycles) -> 2 units
#   hi planes alone    (the attention's P V product with f32_precision "fast")                                                 -> 1 unit
# That derating explains the gap; it is not a peak: it goes into `issued_units_per_product` / `frac_of_issue_bound` (= frac x units) beside `frac`.
# For context the exact-fp32 matrix peak (v_mfma_f32_*_f32 / xf32-less gfx950: 157.3 TFLOP/s) is printed as `exact_fp32_matrix_peak`.
HW_MFMA_PEAK_TFLOPS = 2500.0
EXACT_FP32_MATRIX_PEAK_TFLOPS = 157.3


def f32_units(model):
    """(attention units, GEMM units) per algorithmic product of the fp32 mode for the LAST forward of `model`: the operand form it actually took
    (option "last_f32_form", set by rz_vision_forward — bench.py no longer re-derives the padding / form rules) and the options in force."""
    form, mxa, pv = model.get_model_option("last_f32_form"), model.get_model_option("attn_f32_mx"), model.get_model_option("attn_f32_pv")
    if form == 0:
        return None                     # exact-fp32 MFMA kernels (both split switches off, or a weight beyond the f16 range)
    mx_form = form == 2
    gemm = 2.0 if mx_form else 3.0
    scores = 2.0 if (mx_form and mxa >= 2) else 3.0
    pv_units = 1.0 if pv else (2.0 if (mx_form and mxa >= 1) else 3.0)
    return (scores + pv_units) / 2.0, gemm


def issue_units(dtype, model, cfg, side, prompts):
    """None for the 16-bit modes; for the fp32 mode {"attn", "gemm", "whole", "form"}: MFMA units issued per algorithmic product (call AFTER a forward)."""
    if dtype != "f32":
        return None
    u = f32_units(model)
    if u is None:
        return {"attn": None, "gemm": None, "whole": None, "form": "exact-fp32 MFMA kernels"}
    ua, ug = u
    f_img = flops_per_image(cfg, side, prompts)
    f_attn = cfg.num_blocks * attention_flops_per_image_layer(cfg, side)
    return {"attn": ua, "gemm": ug, "whole": round((f_attn * ua + (f_img - f_attn) * ug) / f_img, 4),
            "form": {1: "three f16 planes", 2: "MX form (f16 hi plane + block-scaled e4m3 correction planes)"}[model.get_model_option("last_f32_form")],
            "npad": model.get_model_option("last_npad")}


def roofline_mfma(kernel, achieved_tflops, units, extra):
    """One MFMA-bound roofline block: `frac` against the hardware peak; the fp32 mode adds the issue-bound view."""
    r = {"kernel": kernel, "bound": "mfma", "achieved": round(achieved_tflops, 2), "peak": HW_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
         "frac": round(achieved_tflops / HW_MFMA_PEAK_TFLOPS, 4)}
    r.update(extra)
    if units is not None:
        r["issued_units_per_product"] = units
        r["frac_of_issue_bound"] = round(achieved_tflops * units / HW_MFMA_PEAK_TFLOPS, 4)
        r["issue_bound_note"] = ("fp32 mode: each algorithmic product issues `issued_units_per_product` f16-MFMA units (split operands, DESIGN.md §4.4); "
                                 "frac_of_issue_bound = frac x units is the share of the matrix pipe's issue slots in use — an explanation of the gap, not a peak")
        r["exact_fp32_matrix_peak"] = EXACT_FP32_MATRIX_PEAK_TFLOPS
        r["vs_exact_fp32_matrix_peak"] = round(achieved_tflops / EXACT_FP32_MATRIX_PEAK_TFLOPS, 3)
    return r


DTYPES = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}


def pmc_traffic(kernel, batch, side, dtype):
    """HBM bytes per launch of `kernel` from the committed PMC passes (a separate rocprofv3 --pmc run cannot happen inside
    this process); null unless the passes were taken on exactly this workload.  -> (bytes, "profiles/<round>/<file>")"""
    fname = "hbm_traffic_pmc_f32.json" if dtype == "f32" else "hbm_traffic_pmc.json"
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", rnd, fname)))
            c = rec["config"]
            if (c["batch"], c["image_side"], c["dtype"]) == (batch, side, dtype):
                return rec["kernels"][kernel]["hbm_bytes_per_launch"], f"profiles/{rnd}/{fname}"
        except (OSError, KeyError, ValueError):
            pass
    return None, None


def physical_cores():
    """Physical cores of the host: distinct (physical id, core id) pairs of /proc/cpuinfo (SMT siblings share one); logical count if unreadable."""
    try:
        seen, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        if phys is not None and core is not None:
            seen.add((phys, core))
        if seen:
            return len(seen)
    except OSError:
        pass
    return os.cpu_count() or 1


def recorded_thread_sweep():
    """profiles/r06/cpu_threads_sweep.json (tools/cpu_threads_sweep.py, run once on the GPU box's host): images/s of this CPU leg at 16 / 32 / 64 / 128 threads."""
    for rnd in ("r06",):
        try:
            return json.load(open(os.path.join(ROOT, "profiles", rnd, "cpu_threads_sweep.json"))), f"profiles/{rnd}/cpu_threads_sweep.json"
        except (OSError, ValueError):
            pass
    return None, None


def cpu_limits():
    """(logical CPUs of the host, CPUs in this process's affinity mask, cgroup CPU quota in CPUs or None)"""
    host = os.cpu_count() or 1
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = host
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(int(q) / int(period)))
    except (OSError, ValueError):
        pass
    return host, aff, quota


def usable_cores():
    """Threads of the CPU baseline: what this process may use (affinity mask, cgroup quota); on an unconstrained view of a big host the thread count that
    MEASURED fastest in the recorded sweep (a measurement, not a choice), 16 if no sweep is committed."""
    host, aff, quota = cpu_limits()
    n = min(host, aff, quota if quota is not None else host)
    if n > 64:
        sweep, _ = recorded_thread_sweep()
        n = min(n, int(sweep["fastest_threads"])) if sweep else 16
    return n


def cores_note():
    """Where the thread count of the CPU baseline comes from (BASELINE.md §3 asks for the core count to be stated)."""
    host, aff, quota = cpu_limits()
    used = usable_cores()
    sweep, src = recorded_thread_sweep()
    sweep_txt = ("; recorded thread sweep on this CPU model (" + src + "): " + ", ".join(f"{k} threads {v} images/s" for k, v in sweep["images_per_s_by_threads"].items())) if sweep else ""
    if quota is not None and used == quota:
        why = f"the cgroup's CPU quota ({quota} CPUs: more threads only oversubscribe it{sweep_txt})"
    elif used == min(host, aff):
        why = "all of them"
    elif sweep:
        why = "the fastest of the recorded sweep" + sweep_txt
    else:
        why = "no thread sweep is committed: one GPU's share of an 8-GPU host"
    return (f"host exposes {host} logical CPUs = {physical_cores()} physical cores, affinity mask {aff}, cgroup CPU quota {quota if quota is not None else 'none'}; "
            f"{used} threads used ({why})")


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


# last recorded figures of the two optional CPU legs (BENCH_r04, 2026-10: AMD EPYC 9575F, the driver's own run of this file): kept in `sample` so that
# the default run stays short (83.5 s of a 289 s run were one all-256-CPU pass)
CPU_RECORDED = ("recorded earlier (BENCH_r04, round 4, same CPU model): eager attention (what the pinned transformers 4.39.3 runs; materialises the 12 x N x N "
                "scores) 0.0915 images/s on 16 threads (10.9 s per image); SDPA on ALL 256 logical CPUs of the host 0.0120 images/s (83.5 s per image: the other "
                "cores belong to the other GPUs' jobs) — re-measure with --cpu-eager / --cpu-all-cores")


def cpu_baseline(cfg, sd, side, n_prompts, ids, mask, eager=False, all_cores=False):
    """Oracle (CPU port of the reference path, oracle/radzero_oracle.py) timed on this host's cores on a bounded sample of
    the same workload: prompts encoded once, one warm-up pass, then the MEDIAN of 3 timed passes of 2 images with SDPA
    attention (transformers-5 default; the faster path on every box measured, hence the denominator of any speed-up claim) — about 20 s.
    `eager` adds the pinned transformers 4.39.3's attention (1 image per pass, ~45 s), `all_cores` one pass on every logical CPU of the
    host (~85 s); without them their last recorded figures are quoted in `sample`."""
    from oracle.radzero_oracle import OracleModel      # baseline leg only; never on the product path
    from radzero_amd.synthetic import synthetic_pixels
    cores = usable_cores()
    torch.set_num_threads(cores)
    enc = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
    res = {}
    for impl, nimg in (("sdpa", 2),) + ((("eager", 1),) if eager else ()):
        om = OracleModel(sd, cfg, attn_impl=impl)
        px = torch.from_numpy(synthetic_pixels(nimg, side, 1234))
        with torch.no_grad():
            tf = om.text_features(enc, split_rows=False)
            om.compute_logits(px[:1, :, : side // 2, : side // 2], [enc], text_features=tf)   # warm-up (thread pool, allocator)
            times = []
            for _ in range(3):
                t0 = time.time()
                om.compute_logits(px, [enc], text_features=tf)
                times.append(time.time() - t0)
        med = sorted(times)[1]
        res[impl] = (nimg / med, nimg, med)
    best = max(res, key=lambda k: res[k][0])
    desc = "; ".join(f"{k}: {v[0]:.4f} images/s (median of 3 x {v[1]} image(s), {v[2]:.1f} s per pass)" for k, v in res.items())
    all_cpus = os.cpu_count() or cores
    all_note = ""
    if all_cores and all_cpus > cores:
        try:
            torch.set_num_threads(all_cpus)
            om = OracleModel(sd, cfg, attn_impl=best)
            px = torch.from_numpy(synthetic_pixels(1, side, 1234))
            with torch.no_grad():
                tf = om.text_features(enc, split_rows=False)
                om.compute_logits(px[:1, :, : side // 2, : side // 2], [enc], text_features=tf)
                t0 = time.time()
                om.compute_logits(px, [enc], text_features=tf)
                dt_all = time.time() - t0
            all_note = f"; ALL {all_cpus} logical CPUs of the host, {best}: {1 / dt_all:.4f} images/s (one pass of 1 image, {dt_all:.1f} s)"
        except Exception as e:           # never lose the bench line to the optional leg
            all_note = f"; all-{all_cpus}-CPU pass failed: {type(e).__name__}"
        finally:
            torch.set_num_threads(cores)
    recorded = "" if (eager and all_cores) else "; " + CPU_RECORDED
    return {"value": round(res[best][0], 5), "unit": "images/s", "cores": cores, "physical_cores_of_host": physical_cores(), "kind": "port",
            "sample": f"{side}x{side} x {n_prompts} cached prompts, fp32, torch CPU, {cores} threads on {cpu_model_name()} [{cores_note()}], "
                      f"1 warm-up + median of 3; {desc}; value = {best}{all_note}{recorded}"}


def node_shared_state_dict(cfg, seed, local_rank, multi):
    """The synthetic checkpoint (840 MB of numpy Philox) is generated ONCE per node: local rank 0 writes it as model.safetensors
    (temporary name + atomic rename), the other ranks wait at a barrier and read the file — 1 s each instead of N generators
    running side by side before the first barrier."""
    if not multi:
        return make_state_dict(cfg, seed)
    import tempfile
    from radzero_amd.checkpoint import load_checkpoint, save_checkpoint
    d = os.path.join(tempfile.gettempdir(), f"radzero_bench_ckpt_{seed}_{os.environ.get('MASTER_PORT', '0')}")
    f = os.path.join(d, "model.safetensors")
    sd = None
    if local_rank == 0:
        sd = make_state_dict(cfg, seed)
        tmp = d + f".tmp{os.getpid()}"
        save_checkpoint(sd, tmp)
        os.makedirs(d, exist_ok=True)
        os.replace(os.path.join(tmp, "model.safetensors"), f)
        os.rmdir(tmp)
    dist.barrier()
    if sd is None:
        sd = load_checkpoint(f)
    dist.barrier()
    if local_rank == 0:
        try:
            os.remove(f)
            os.rmdir(d)
        except OSError:
            pass
    return sd


def workload_label(B, S, T, dtype, maps, n_tok):
    """Which BASELINE.json config a (batch, side, prompts, dtype, maps) tuple is — never a fixed string."""
    if (B, S, T, dtype, maps) == (32, 1024, 14, "bf16", "none"):
        tag = "BASELINE configs[1] (per-GPU shape of configs[2])"
    elif (B, S, T, dtype) == (16, 1024, 64, "bf16") and maps != "none":
        tag = "BASELINE configs[3] (grounding, per-pixel maps)" if maps == "upsample" else "BASELINE configs[3] variant (fused grounding points instead of maps)"
    elif (S, T, dtype) == (1536, 193, "f16"):
        tag = "BASELINE configs[4] per-GPU shape (8 images over 8 GPUs)" if B == 1 else f"BASELINE configs[4] shape at batch {B}/GPU"
    elif (B, S, T, dtype) == (16, 1024, 64, "f32") and maps == "upsample":
        tag = "BASELINE configs[3] shape in f32 (the 1e-3 mode; configs[3] itself is bf16)"
    elif (B, S, T, dtype) == (1, 1536, 193, "f32"):
        tag = "BASELINE configs[4] per-GPU shape in f32 (the 1e-3 mode; configs[4] itself is fp16)"
    elif (B, S, T, maps) == (32, 1024, 14, "none"):
        tag = f"BASELINE configs[1] shape in {dtype} (configs[1] itself is bf16)"
    elif (B, S, T, maps) == (64, 518, 14, "none"):
        tag = "the reference's own operating point (img_size 518, eval batch 64: exp/cxr_pt/configs/radzero.yaml:19, config.yaml:55; not a BASELINE config)"
    else:
        tag = "custom shape (not a BASELINE config)"
    return (f"{tag}: batch={B}/GPU {S}x{S} synthetic CXR, {T} prompts, N={n_tok} tokens/image, 12 ViT + 2 align blocks, "
            f"VL-CABS head; text embeddings cached; maps={maps}")


class InputPipeline:
    """Where a step's pixel_values come from, beyond "already resident in HBM" (never the headline number):
      host   : fp32 pixels in PINNED host memory, H2D on a copy stream into a double buffer, overlapped with the previous step's forward
      raw    : raw uint16 2048x1760 detector images resident in HBM -> batched device preprocessing (radzero_amd/preprocess.py: min-max
               to 8 bit, Pillow-exact bicubic to S x S, rescale, normalise) on a side stream into a double buffer, overlapped likewise
      rawhost: the same raw images in pinned host memory (7.2 MB each instead of 12.6 MB of fp32 pixels): H2D + preprocessing overlapped
    `next()` returns the pixel tensor for this step and starts producing the next step's."""

    def __init__(self, mode, pixels, device, overlap=True):
        self.mode, self.device, self.overlap = mode, device, overlap
        B, _, S, _ = pixels.shape
        self.main = torch.cuda.current_stream(device)
        self.side = torch.cuda.Stream(device=device) if overlap else self.main
        self.buf = [torch.empty_like(pixels), torch.empty_like(pixels)]
        self.ready = [torch.cuda.Event(), torch.cuda.Event()]
        self.consumed = [torch.cuda.Event(), torch.cuda.Event()]
        self.k = 0
        if mode == "host":
            self.src = pixels.cpu().pin_memory()
        else:
            import numpy as np
            from radzero_amd.preprocess import DevicePreprocessor
            from radzero_amd.synthetic import synthetic_cxr_raw
            self.pre = DevicePreprocessor(S, device=device)
            raws = [torch.from_numpy(synthetic_cxr_raw("uint16", (2048, 1760), 900 + i).view(np.uint16)) for i in range(min(B, 4))]
            raws = [raws[i % len(raws)] for i in range(B)]
            self.raw_host = [r.pin_memory() for r in raws] if mode == "rawhost" else None
            self.raw_dev = [r.to(device) for r in raws]
        for e in self.consumed:
            e.record(self.main)
        self._produce(0)

    def _produce(self, j):
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.consumed[j])            # the forward that read this buffer two steps ago has finished
            if self.mode == "host":
                self.buf[j].copy_(self.src, non_blocking=True)
            else:
                raws = [r.to(self.device, non_blocking=True) for r in self.raw_host] if self.raw_host is not None else self.raw_dev
                self.pre(raws, out=self.buf[j])
            self.ready[j].record(self.side)

    def next(self):
        j = self.k & 1
        self.main.wait_event(self.ready[j])
        if self.overlap:
            self._produce(j ^ 1)                               # runs beside this step's forward
        self.k += 1
        return self.buf[j], j

    def done(self, j):
        self.consumed[j].record(self.main)
        if not self.overlap:
            self._produce(j ^ 1)


def short_run(sd, cfg, device, dtype, B, S, T, maps, min_len, max_len, steps=5, warmup=2, pipeline=None, f32_precision=None):
    """A few steps of another BASELINE config inside the same process (rank 0, N=1 only), so that the driver's clock and
    the JSON line cover it: same timed-region rules as the main workload.  Every entry carries its own roofline block for the
    kernel that dominates it (attention, MFMA-bound) and, with per-pixel maps, for the HBM-bound upsampling kernel."""
    from radzero_amd.modeling import RadZeroModel
    model = RadZeroModel.from_state_dict(sd, cfg, torch_dtype=DTYPES[dtype], device=device).eval()
    try:
        if f32_precision:
            model.set_f32_precision(f32_precision)
        g = torch.Generator(device=device).manual_seed(4242)
        px = torch.randn((B, 3, S, S), generator=g, device=device, dtype=torch.float32)
        ids, mask = synthetic_prompts(T, min_len, max_len, 4321)
        enc = {"input_ids": torch.from_numpy(ids).to(device), "attention_mask": torch.from_numpy(mask).to(device)}
        tf = model.forward_text_model(enc)["text_features_wo_l2_norm"]
        pipe = InputPipeline(pipeline, px, device) if pipeline else None
        # per-pixel maps land in ONE buffer re-used by every step (4.29 GB at cfg 4): a fresh torch.empty per step put a hipMalloc of
        # that size inside some timed regions (325 against 402 images/s between two runs of the same tree)
        maps_buf = torch.empty((B * T, S, S), dtype=torch.float32, device=device) if maps == "upsample" else None

        def step():
            if pipe is not None:
                pxs, j = pipe.next()
                out = model.compute_logits(pxs, [enc], text_features=tf)
                pipe.done(j)
                return out
            out = model.compute_logits(px, [enc], text_features=tf)
            if maps == "upsample":
                out["similarity_maps"] = model.upsample_similarity(out["similarity_scores"], (S, S), out=maps_buf)
            elif maps == "points":
                out["grounding_points"] = model.grounding_points(out["similarity_scores"], (S, S))
            return out

        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        model.profile(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        prof = model.profile_read()
        model.profile(False)
        assert bool(torch.isfinite(out["logits"]).all())
        ips = B * steps / dt
        graph_ips = None
        if B == 1 and maps == "none":
            # latency-bound shape: the same step replayed as ONE hipGraph (RadZeroModel.make_graphed), pixels copied into the
            # graph's static input inside the timed region
            run = model.make_graphed(px.shape, [enc])
            for _ in range(warmup):
                run(px)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                gout = run(px)
            torch.cuda.synchronize()
            graph_ips = B * steps / (time.perf_counter() - t0)
            assert bool(torch.isfinite(gout["logits"]).all())
        f_img = flops_per_image(cfg, S, T)
        iu = issue_units(dtype, model, cfg, S, T)
        guard_reruns = model.guard_reruns() if dtype == "f32" else None
        attn_ms = prof["attn"]["ms"] / max(1, prof["attn"]["launches"])
        attn_tf = B * attention_flops_per_image_layer(cfg, S) / (attn_ms * 1e-3) / 1e12 if attn_ms > 0 else None
        roof = None
        if attn_tf is not None:
            kname = "flash_attn_split_kernel" if dtype == "f32" else "flash_attn_kernel"
            traffic, traffic_src = pmc_traffic(kname, B, S, dtype)
            roof = roofline_mfma(kname, attn_tf, iu["attn"] if iu else None,
                                 {"traffic": traffic, **({"traffic_unit": f"HBM bytes per launch, recorded by separate rocprofv3 --pmc passes ({traffic_src}), not measured in this run"} if traffic is not None else {}),
                                  "avg_launch_ms": round(attn_ms, 4), "launches": prof["attn"]["launches"]})
        roof_post = None
        if maps == "upsample" and prof["post"]["launches"] > 0:
            # upsample_bilinear_kernel: algorithmic bytes = the fp32 maps it writes, B*T*S*S*4 (the patch-grid input is 0.5 % of that)
            up_ms = prof["post"]["ms"] / prof["post"]["launches"]
            gbs = B * T * S * S * 4 / (up_ms * 1e-3) / 1e9
            roof_post = {"kernel": "upsample_bilinear_kernel", "bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(gbs / 8000.0, 4), "traffic": None, "algorithmic_bytes": int(B * T * S * S * 4),
                         "avg_launch_ms": round(up_ms, 4), "launches": prof["post"]["launches"]}
        label = workload_label(B, S, T, dtype, maps, cfg.tokens(S))
        if f32_precision:
            label += f"; f32_precision={f32_precision}"
        if pipeline:
            label += {"host": "; INPUT = fp32 pixels from pinned host memory every step, H2D on a copy stream into a double buffer (overlapped)",
                      "raw": "; INPUT = raw uint16 2048x1760 images resident in HBM -> batched device preprocessing on a side stream (overlapped)",
                      "rawhost": "; INPUT = raw uint16 2048x1760 images in pinned host memory -> H2D + batched device preprocessing on a side stream (overlapped)"}[pipeline]
        return {"workload": label, "dtype": dtype, "steps": steps, "warmup": warmup,
                **({"roofline": roof} if roof else {}), **({"roofline_upsample": roof_post} if roof_post else {}),
                "images_per_s": round(ips, 3), "similarity_maps_per_s": round(ips * T, 2), "ms_per_step": round(dt / steps * 1e3, 3),
                "model_tflops_per_s": round(ips * f_img / 1e12, 2),
                "frac_of_mfma_peak_whole_path": round(ips * f_img / 1e12 / HW_MFMA_PEAK_TFLOPS, 4),
                **({"issued_units_per_product_whole_path": iu["whole"], "f32_operand_form": iu["form"],
                    "frac_of_issue_bound_whole_path": round(ips * f_img / 1e12 * iu["whole"] / HW_MFMA_PEAK_TFLOPS, 4)} if iu and iu["whole"] else {}),
                "attention_tflops_per_s": None if attn_tf is None else round(attn_tf, 1),
                **({} if guard_reruns is None else {"f32_split_guard_reruns": guard_reruns}),
                **({} if graph_ips is None else {"images_per_s_hipgraph_replay": round(graph_ips, 3)}),
                "kernel_family_ms_per_step": {k: round(v["ms"] / steps, 3) for k, v in prof.items()}}
    finally:
        model.close()
        del model
        torch.cuda.empty_cache()


def driver_leg(sd, cfg, device, dtype="bf16", B=64, S=518, T=14, n_batches=24):
    """The reference's evaluation loop at its own operating point (calculate_similarities over a DataLoader of 518^2 x 64 batches:
    exp/cxr_pt/inference/utils.py:70-106, configs/radzero.yaml:19, config.yaml:55) through radzero_amd.inference.calculate_similarities, with and
    without batch shaping (the stream of 64-image batches re-cut into forwards of model.preferred_batch(64, 518, 518) = 62 images: whole rounds
    of the persistent GEMM's 256 tiles).  Pixels resident in HBM; timed: the whole driver call, logits on the host at the end."""
    from radzero_amd.inference import calculate_similarities
    from radzero_amd.modeling import RadZeroModel
    model = RadZeroModel.from_state_dict(sd, cfg, torch_dtype=DTYPES[dtype], device=device).eval()
    try:
        g = torch.Generator(device=device).manual_seed(99)
        px = torch.randn((B, 3, S, S), generator=g, device=device, dtype=torch.float32)
        ids, mask = synthetic_prompts(T, 6, 10, 4321)
        tb = {"encoded_key_phrases": {"input_ids": torch.from_numpy(ids).to(device), "attention_mask": torch.from_numpy(mask).to(device)}}
        res, logits = {}, {}
        for shaping in (False, True, False, True):
            calculate_similarities([px] * 2, tb, model, batch_shaping=shaping)            # warm-up: workspaces for this forward size
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            logits[shaping] = calculate_similarities([px] * n_batches, tb, model, batch_shaping=shaping)
            dt = time.perf_counter() - t0
            res.setdefault(shaping, []).append(B * n_batches / dt)
        same = bool((logits[True] == logits[False]).all())
        pref = model.preferred_batch(B, S, S)
        return {"workload": f"batch driver calculate_similarities over {n_batches} batches of {B} x {S}x{S} images, {T} cached prompts, {dtype}; pixels resident in HBM, "
                            f"logits to the host at the end (the reference's evaluation loop at its own operating point)",
                "images_per_s_as_batched": round(max(res[False]), 1), "images_per_s_batch_shaping": round(max(res[True]), 1),
                "forward_size_with_shaping": pref, "gain": round(max(res[True]) / max(res[False]), 4), "logits_bit_identical": same,
                "all_runs_images_per_s": {"as_batched": [round(v, 1) for v in res[False]], "batch_shaping": [round(v, 1) for v in res[True]]}}
    finally:
        model.close()
        del model
        torch.cuda.empty_cache()


def request_leg(sd, cfg, device, dtype="bf16", S=1024, steps=24, warmup=6):
    """The reference's per-request path (eval_refer_grounding, exp/cxr_pt/inference/grounding_utils.py:283-326; extract_similarity_map,
    visualization/attention_map_base.py:12-42): ONE image and ONE text that the model has never seen per request, compute_logits + the
    fused grounding point, the point read back on the host before the next request starts (the reference does unravel_index on the
    host).  No text cache hit is possible: every request carries a fresh token row.  Reported: ms per request eager (text encoder on a
    side stream beside the vision forward), the same with the text embedding supplied (=> the text encoder's share of the latency),
    the text encoder alone, and the whole request replayed as ONE hipGraph."""
    from radzero_amd.modeling import RadZeroModel
    model = RadZeroModel.from_state_dict(sd, cfg, torch_dtype=DTYPES[dtype], device=device).eval()
    try:
        g = torch.Generator(device=device).manual_seed(777)
        px = torch.randn((1, 3, S, S), generator=g, device=device, dtype=torch.float32)
        L = 12
        encs = []
        for i in range(steps + warmup):                      # tokenizer(text, padding=True) of one text: no pad tokens, 8-12 tokens
            ids, mask = synthetic_prompts(1, 8, L, 9000 + i)
            n = int(mask.sum())
            encs.append({"input_ids": torch.from_numpy(ids[:, :n].copy()).to(device), "attention_mask": torch.from_numpy(mask[:, :n].copy()).to(device)})
        fixed = [synthetic_prompts(1, L, L, 9500 + i) for i in range(steps + warmup)]      # the graph leg: a fixed token count

        def request(enc, feats=None):
            out = model.compute_logits(px, [enc], text_features=feats)
            return model.grounding_points(out["similarity_scores"], (S, S)).cpu()          # host sync: one per request, as in the reference

        def timed(fn, n0, n1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(n0, n1):
                fn(k)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / (n1 - n0) * 1e3

        for ln in range(8, L + 1):                            # relative-position tables of every length that will occur
            ids, mask = synthetic_prompts(1, ln, ln, 1)
            model.forward_text_model({"input_ids": torch.from_numpy(ids).to(device), "attention_mask": torch.from_numpy(mask).to(device)})
        for k in range(warmup):
            request(encs[k])
        eager = timed(lambda k: request(encs[k]), warmup, warmup + steps)
        feats0 = model.forward_text_model(encs[0])["text_features_wo_l2_norm"]
        cached = timed(lambda k: request(encs[k], feats0), warmup, warmup + steps)
        model.text_cache_enabled = False
        text_alone = timed(lambda k: model.forward_text_model(encs[k])["text_features_wo_l2_norm"].sum().item(), warmup, warmup + steps)
        model.text_cache_enabled = True
        run = model.make_graphed_request(px.shape, (1, L), points=True)
        dev_fixed = [(torch.from_numpy(i).to(device), torch.from_numpy(m).to(device)) for i, m in fixed]
        for k in range(warmup):
            run(px, *dev_fixed[k])["grounding_points"].cpu()
        graph = timed(lambda k: run(px, *dev_fixed[k])["grounding_points"].cpu(), warmup, warmup + steps)
        return {"workload": f"per-request refer-grounding (grounding_utils.py:283-326): batch 1, {S}x{S}, ONE text never seen before per request "
                            f"(8-{L} tokens, no cache hit), fused grounding point read on the host every request; {dtype}",
                "ms_per_request": round(eager, 3), "requests_per_s": round(1e3 / eager, 1),
                "ms_per_request_text_embedding_supplied": round(cached, 3), "text_encoder_share_ms": round(eager - cached, 3),
                "text_encoder_alone_ms": round(text_alone, 3), "ms_per_request_hipgraph": round(graph, 3), "steps": steps}
    finally:
        model.close()
        del model
        torch.cuda.empty_cache()


def text_encode_steady(model, device, sizes=((14, 6, 10), (64, 8, 32), (193, 6, 16)), reps=5):
    """Steady-state cost of encoding a NEW prompt set of T prompts (second and later calls: workspaces, relative-position table and the
    library's buffers exist) — the number beside the cold `text_encode_once_ms`.  Median of `reps` calls, each with fresh token ids."""
    out = {}
    was = model.text_cache_enabled
    model.text_cache_enabled = False
    try:
        for T, lo, hi in sizes:
            ts = []
            for r in range(reps + 1):
                ids, mask = synthetic_prompts(T, lo, hi, 5000 + 31 * T + r)
                enc = {"input_ids": torch.from_numpy(ids).to(device), "attention_mask": torch.from_numpy(mask).to(device)}
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                model.forward_text_model(enc)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            out[str(T)] = round(sorted(ts[1:])[len(ts[1:]) // 2], 3)
    finally:
        model.text_cache_enabled = was
    return out


class CollectiveLog:
    """What torch.distributed actually carried, read from the live process group: every collective entry point this process can
    reach is wrapped with a counter (name, payload bytes) for the lifetime of the bench, and the log is cut into phases — the one-time
    prompt exchange, the timed region's steps, the contract's barriers."""
    NAMES = ("all_gather_into_tensor", "all_gather", "all_reduce", "gather", "broadcast", "reduce_scatter_tensor", "all_to_all_single",
             "all_gather_object", "barrier")

    def __init__(self):
        self.calls, self.phase = [], "setup"
        for name in self.NAMES:
            fn = getattr(dist, name, None)
            if fn is None:
                continue
            setattr(dist, name, self._wrap(name, fn))

    def _wrap(self, name, fn):
        def wrapped(*a, **k):
            nbytes = max((int(t.numel() * t.element_size()) for t in a if torch.is_tensor(t)), default=0)     # the gathered / reduced buffer
            self.calls.append((self.phase, name, nbytes))
            return fn(*a, **k)
        return wrapped

    def count(self, phase, exclude=("barrier",)):
        return sum(1 for ph, n, _ in self.calls if ph == phase and n not in exclude)

    def bytes(self, phase, name):
        return sum(b for ph, n, b in self.calls if ph == phase and n == name)


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def visible_gpu_count():
    """GPUs this node exposes to this process, WITHOUT loading any GPU library (torch.cuda.device_count() falls back to hipGetDeviceCount — which
    initialises HIP / HSA in the parent — when the amdsmi module is missing): KFD topology nodes with simd_count > 0, narrowed by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES.  None when the topology is not readable (then the children decide)."""
    import glob
    n = 0
    try:
        nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
        if not nodes:
            return None
        for path in nodes:
            for line in open(path):
                k, _, v = line.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n += 1
    except (OSError, ValueError):
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(n, argv, child_cmd=None, grace_s=30.0, overall_timeout_s=None):
    """`python bench.py --gpus N` without a launcher around it: start N FRESH child processes, one rank per GPU (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment — what torch.distributed.run would set), relay rank 0's JSON line, return the children's
    worst exit code.  This parent never touches the GPU (no torch.cuda call, no HIP library load): a process that has initialised the GPU
    must not fork / exec workers on this pool.  A failed rank is reported, never retried; once one rank has died the others get `grace_s`
    to exit by themselves (their barrier will never complete) and are then terminated by PID.
    `child_cmd` (tests: RZ_BENCH_CHILD_CMD, a JSON list) replaces `[sys.executable, bench.py]`.  `overall_timeout_s` (default
    RZ_BENCH_LAUNCH_TIMEOUT_S or 3600): ranks that all hang without any of them exiting are terminated by PID instead of waited for forever."""
    import subprocess
    import threading
    cmd = list(child_cmd) if child_cmd else [sys.executable, os.path.abspath(__file__)]
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs, lines = [], []

    def relay(rank, pipe):
        for raw in pipe:
            line = raw.rstrip("\n")
            if rank == 0 and line.startswith("{") and '"metric"' in line:
                lines.append(line)
            else:
                print(f"[rank {rank}] {line}", file=sys.stderr, flush=True)

    threads = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), GROUP_RANK="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   RZ_BENCH_LAUNCHED_BY="bench.py")
        p = subprocess.Popen(cmd + list(argv), env=env, stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1)
        t = threading.Thread(target=relay, args=(r, p.stdout), daemon=True)
        t.start()
        procs.append(p)
        threads.append(t)
    codes = [None] * n
    failed_at = None
    if overall_timeout_s is None:
        overall_timeout_s = float(os.environ.get("RZ_BENCH_LAUNCH_TIMEOUT_S", "3600"))
    started = time.time()
    while any(c is None for c in codes):
        if failed_at is None and time.time() - started > overall_timeout_s:
            print(f"[bench launcher] no rank has finished after {overall_timeout_s:.0f} s: terminating all ranks", file=sys.stderr, flush=True)
            failed_at = time.time() - grace_s - 1.0
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
                if codes[r] not in (None, 0) and failed_at is None:
                    failed_at = time.time()
                    print(f"[bench launcher] rank {r} exited with code {codes[r]}", file=sys.stderr, flush=True)
        if failed_at is not None and time.time() - failed_at > grace_s:
            for r, p in enumerate(procs):
                if codes[r] is None:
                    print(f"[bench launcher] terminating rank {r} (pid {p.pid}): another rank failed", file=sys.stderr, flush=True)
                    p.terminate()
                    try:
                        p.wait(timeout=10)
                    except subprocess.TimeoutExpired:
                        p.kill()
                    codes[r] = p.wait()
        time.sleep(0.05)
    for t in threads:
        t.join(timeout=5)
    worst = max((abs(c) for c in codes), default=0)
    if worst == 0 and len(lines) != 1:
        print(f"[bench launcher] expected ONE JSON line from rank 0, got {len(lines)}", file=sys.stderr, flush=True)
        worst = 1
    if worst == 0:
        print(lines[0], flush=True)
    else:
        print(f"[bench launcher] exit codes by rank: {codes}", file=sys.stderr, flush=True)
    return min(worst, 255)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--side", type=int, default=1024)
    ap.add_argument("--prompts", type=int, default=14)
    ap.add_argument("--dtype", default="bf16", choices=list(DTYPES))
    ap.add_argument("--f32-precision", default=None, choices=["high", "fast"], help="fp32 mode only: RadZeroModel.set_f32_precision (default high)")
    ap.add_argument("--maps", default="none", choices=["none", "upsample", "points"],
                    help="similarity-map post-processing inside the step (BASELINE cfg 4): per-pixel bilinear maps or fused grounding points")
    ap.add_argument("--min-len", type=int, default=6)
    ap.add_argument("--max-len", type=int, default=10)
    ap.add_argument("--force-dist", action="store_true", help="rehearsal: initialise the RCCL process group even with one rank")
    ap.add_argument("--launch", action="store_true", help="start the rank(s) as fresh child processes even for --gpus 1 (rehearsal of the N > 1 launcher on a one-GPU box; implied by --gpus N > 1 outside torch.distributed.run)")
    ap.add_argument("--host-pixels", action="store_true", help="measurement only (never the headline): fp32 pixels start in pinned host memory and cross PCIe inside every step, on a copy stream into a double buffer (overlapped with the previous step)")
    ap.add_argument("--raw-images", default=None, choices=["device", "host"], help="measurement only (never the headline): every step starts from raw uint16 2048x1760 images (resident in HBM / in pinned host memory) and runs the batched device preprocessing on a side stream")
    ap.add_argument("--no-overlap", action="store_true", help="A/B of the two input modes above: produce the inputs on the compute stream (the un-overlapped behaviour of round 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-eager", action="store_true", help="CPU baseline: also time eager attention (the pinned transformers 4.39.3's; ~45 s more)")
    ap.add_argument("--cpu-all-cores", action="store_true", help="CPU baseline: also one pass on every logical CPU of the host (~85 s more)")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--all-kernel-events", action="store_true", help="A/B: record HIP events for every kernel family inside the timed region (default: "
                    "the dominant kernel's family only; the other families are timed over two extra, un-timed steps)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of BASELINE configs[3], configs[4] per-GPU shape and the fp32 mode")
    ap.add_argument("--attn-variant", type=int, default=None, help="A/B switch (rz_set_option): 0 = default (4 waves x 32 query rows; bf16 without the running maximum), 417 = running maximum tracked; RZ_EXPERIMENTS=1 library: 64 / 128 / ...")
    ap.add_argument("--ln-fused", type=int, default=None, help="A/B switch: 1 fused LayerNorm (default, 16-bit modes), 0 stand-alone LayerNorm kernels")
    ap.add_argument("--gemm-variant", type=int, default=None, help="A/B switch: 0 auto, 1 / 3 / 7 / 8 force that GEMM kernel (include/radzero_hip.h); RZ_EXPERIMENTS=1 library: 10 / 11 / 12")
    args = ap.parse_args()

    if (args.gpus > 1 or args.launch) and "WORLD_SIZE" not in os.environ:
        # not under torch.distributed.run: launch the ranks ourselves, before anything in this process touches the GPU
        child = os.environ.get("RZ_BENCH_CHILD_CMD")
        if not child:
            have = visible_gpu_count()                # KFD topology: no GPU library is loaded in this parent
            if have is not None and have < args.gpus:
                raise SystemExit(f"bench.py --gpus {args.gpus}: this node exposes {have} GPU(s)")
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:], json.loads(child) if child else None))

    # HSA reads its environment at hsa_init, i.e. at the first torch.cuda call below: set these before anything touches the GPU
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} under WORLD_SIZE={world}: launch one rank per GPU (or run `python bench.py --gpus N`, which starts them)")
    # RZ_BENCH_DEVICE / RZ_BENCH_BACKEND: test hooks only (tests/test_bench_main_world8_cpu.py rehearses every world > 1 branch of this function on
    # CPU over gloo with a stub model).  The product model has no CPU path: with the real RadZeroModel "cpu" fails at construction.
    dev_kind, backend = os.environ.get("RZ_BENCH_DEVICE", "cuda"), os.environ.get("RZ_BENCH_BACKEND", "nccl")
    if dev_kind == "cuda":
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        sync = torch.cuda.synchronize
    else:
        device = torch.device(dev_kind)
        sync = lambda: None      # noqa: E731
    use_dist = world > 1 or args.force_dist
    clog = None
    if use_dist:
        clog = CollectiveLog()
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **({"device_id": device} if device.type == "cuda" else {}))

    from radzero_amd.modeling import RadZeroModel
    from radzero_amd.parallel import sharded_text_features

    from radzero_amd import _lib
    if args.attn_variant is not None:
        _lib.check(_lib.load().rz_set_option(b"attn_variant", args.attn_variant), "rz_set_option")
    if args.ln_fused is not None:
        _lib.check(_lib.load().rz_set_option(b"ln_fused", args.ln_fused), "rz_set_option")
    if args.gemm_variant is not None:
        _lib.check(_lib.load().rz_set_option(b"gemm_variant", args.gemm_variant), "rz_set_option")

    cfg = RadZeroConfig()
    sd = node_shared_state_dict(cfg, 20260103, local_rank, use_dist and world > 1)
    model = RadZeroModel.from_state_dict(sd, cfg, torch_dtype=DTYPES[args.dtype], device=device).eval()
    if args.f32_precision:
        model.set_f32_precision(args.f32_precision)

    B, S, T = args.batch, args.side, args.prompts
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    pixels = torch.randn((B, 3, S, S), generator=g, device=device, dtype=torch.float32)   # resident in HBM
    ids, mask = synthetic_prompts(T, args.min_len, args.max_len, 4321)
    enc = {"input_ids": torch.from_numpy(ids).to(device), "attention_mask": torch.from_numpy(mask).to(device)}

    # one-time prompt encoding: sharded over ranks + ONE all_gather (RCCL over xGMI), then cached
    sync()
    if clog:
        clog.phase = "prompt_exchange"
    t0 = time.time()
    text_features = sharded_text_features(lambda e: model.forward_text_model(e)["text_features_wo_l2_norm"], enc,
                                          feature_dim=cfg.hidden_size)
    sync()
    text_ms = (time.time() - t0) * 1e3
    if clog:
        clog.phase = "setup"
    text_steady = text_encode_steady(model, device) if (rank == 0 and world == 1) else None

    main_maps_buf = torch.empty((B * T, S, S), dtype=torch.float32, device=device) if args.maps == "upsample" else None
    mode = "host" if args.host_pixels else {None: None, "device": "raw", "host": "rawhost"}[args.raw_images]
    pipe = InputPipeline(mode, pixels, device, overlap=not args.no_overlap) if mode else None

    def step():
        if pipe is not None:
            px, slot = pipe.next()
            out = model.compute_logits(px, [enc], text_features=text_features)
            pipe.done(slot)
        else:
            out = model.compute_logits(pixels, [enc], text_features=text_features)
        if args.maps == "upsample":      # (B, T, S, S) fp32 per-pixel maps (interpolate_similarity_scores semantics), into one re-used buffer
            out["similarity_maps"] = model.upsample_similarity(out["similarity_scores"], (S, S), out=main_maps_buf)
        elif args.maps == "points":      # fused upsample + argmax (get_grounding_point semantics), map never written
            out["grounding_points"] = model.grounding_points(out["similarity_scores"], (S, S))
        return out

    for _ in range(args.warmup):
        step()
    if use_dist:
        dist.barrier()
    sync()
    if not args.no_kernel_events:
        # HIP events (rz_profile_*) on the launch stream around the dominant kernel's launches only: ~110 event pairs per step for every
        # kernel would cost the timed region ~1 %
        model.profile(True, families=None if args.all_kernel_events else ("attn",))
    if clog:
        clog.phase = "timed_steps"
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    own_elapsed = time.perf_counter() - t0          # this rank's K steps, before waiting for the slowest rank
    if clog:
        clog.phase = "closing_barrier"
    if use_dist:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if clog:
        clog.phase = "after"
    prof = None
    fam_steps = args.steps
    if not args.no_kernel_events:
        prof = model.profile_read()
        model.profile(False)
        if not args.all_kernel_events:       # the other families: two extra steps outside the timed region, every family recorded
            model.profile(True)
            for _ in range(2):
                step()
            sync()
            extra = model.profile_read()
            model.profile(False)
            fam_steps = 2
            prof = {k: (prof[k] if k == "attn" else extra[k]) for k in extra}
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    rank_ips = None
    if use_dist:
        mine = torch.tensor([args.batch * args.steps / own_elapsed], dtype=torch.float64, device=device)
        every = torch.empty((world,), dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(every, mine)
        rank_ips = [round(float(v), 3) for v in every.tolist()]
    assert bool(torch.isfinite(out["logits"]).all())

    if rank == 0:
        n_tok = cfg.tokens(S)
        iu = issue_units(args.dtype, model, cfg, S, T)
        total_images = world * B * args.steps
        ips = total_images / elapsed
        f_img = flops_per_image(cfg, S, T)
        res = {
            "metric": f"images/sec, zero-shot classification (vision encoder + VL-CABS), {S}x{S} CXR x {T} prompts",
            "value": round(ips, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workload_label(B, S, T, args.dtype, args.maps, n_tok),
                       "global_batch": world * B, "image_side": S, "n_prompts": T, "parallelism": f"dp{world}",
                       "map_postprocessing": args.maps},
            "similarity_maps_per_s": round(ips * T, 2),
            "model_tflops_per_s": round(ips * f_img / 1e12, 2),
            "frac_of_mfma_peak_whole_path": round(ips * f_img / 1e12 / (HW_MFMA_PEAK_TFLOPS * world), 4),
            **({"issued_units_per_product_whole_path": iu["whole"], "f32_operand_form": iu["form"],
                "frac_of_issue_bound_whole_path": round(ips * f_img / 1e12 * iu["whole"] / (HW_MFMA_PEAK_TFLOPS * world), 4)} if iu and iu["whole"] else {}),
            "text_encode_once_ms": round(text_ms, 2),
        }
        if use_dist:
            # evidence of what the process group saw, read from the live group and the collective log — not declared
            pg_world = dist.get_world_size()
            try:
                rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                rccl_version = None
            res["rccl"] = {"backend": dist.get_backend(), "world_size": pg_world, "rccl_version": rccl_version,
                           "launched_by": os.environ.get("RZ_BENCH_LAUNCHED_BY") or ("torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "environment"),
                           "prompt_exchange_collectives": clog.count("prompt_exchange"),
                           # bytes of the gathered table; the list form (CPU / gloo rehearsal, parallel._all_gather_cpu) logs one rank's buffer
                           "all_gather_bytes": clog.bytes("prompt_exchange", "all_gather_into_tensor") + pg_world * clog.bytes("prompt_exchange", "all_gather"),
                           "all_gather_note": f"ONE all_gather_into_tensor of ({-(-T // pg_world)} x {pg_world} ranks, {cfg.hidden_size}) fp32 prompt embeddings, before the timed region (bytes of the gathered table)",
                           "data_path_collectives": clog.count("timed_steps"),
                           "timed_region_barriers": sum(1 for ph, n, _ in clog.calls if ph == "closing_barrier" and n == "barrier"),
                           "per_rank_images_per_s": {"min": min(rank_ips), "max": max(rank_ips), "all": rank_ips}}
            assert res["n_gpus"] == pg_world
        if text_steady is not None:
            res["text_encode_steady_ms"] = text_steady
            res["text_encode_note"] = ("text_encode_once_ms = the FIRST call of the process (allocations, relative-position table, lazy module load); "
                                       "text_encode_steady_ms[T] = median of 5 later calls with T new prompts each")
        if mode:
            res["config"]["workload"] += {"host": "; PIXELS FROM PINNED HOST MEMORY EVERY STEP (PCIe-inclusive: not the headline number)",
                                          "raw": "; EVERY STEP STARTS FROM RAW uint16 2048x1760 IMAGES IN HBM + device preprocessing (not the headline number)",
                                          "rawhost": "; EVERY STEP STARTS FROM RAW uint16 2048x1760 IMAGES IN PINNED HOST MEMORY + device preprocessing (not the headline number)"}[mode]
            res["config"]["input_overlap"] = not args.no_overlap
        if prof is not None and prof["attn"]["launches"] > 0:
            # dominant kernel: flash attention (54 % of the algorithmic FLOPs at 1024^2).
            # algorithmic FLOPs per launch = B images x 4*N^2*D (QK^T + PV over all 12 heads), SURVEY.md §8(d)
            launches = prof["attn"]["launches"]
            avg_ms = prof["attn"]["ms"] / launches
            flops_launch = args.steps * cfg.num_blocks * B * attention_flops_per_image_layer(cfg, S) / launches
            achieved = flops_launch / (avg_ms * 1e-3) / 1e12
            kname = "flash_attn_split_kernel" if args.dtype == "f32" else "flash_attn_kernel"
            traffic, traffic_src = pmc_traffic(kname, B, S, args.dtype)
            res["roofline"] = roofline_mfma(kname, achieved, iu["attn"] if iu else None, {
                "traffic": traffic,
                "traffic_unit": ("HBM bytes per launch; RECORDED by separate rocprofv3 --pmc passes of this command "
                                 f"(2*FETCH_SIZE + WRITE_SIZE, KiB -> B; {traffic_src}), not measured in this run") if traffic is not None
                                else "no PMC passes are committed for this exact workload (profiles/*/hbm_traffic_pmc*.json cover B = 32, 1024^2, bf16 and f32)",
                "algorithmic_bytes": int(B * cfg.tokens(S) * 768 * (2 if args.dtype != "f32" else 4) * 4),
                "avg_launch_ms": round(avg_ms, 4), "launches": launches})
            res["kernel_family_ms_per_step"] = {k: round(v["ms"] / (args.steps if k == "attn" else fam_steps), 3) for k, v in prof.items()}
            gemm_ms = res["kernel_family_ms_per_step"].get("gemm", 0.0)
            if gemm_ms > 0:
                # the other half of the step: every linear layer of the 14 blocks + the patch embedding (24 N D^2 per block
                # and image, 2 Np 588 D for the conv) over the GEMM family's HIP-event time
                npatch = n_tok - 1
                gemm_flops = B * (cfg.num_blocks * 24.0 * n_tok * cfg.hidden_size ** 2 + 2.0 * npatch * 588 * cfg.hidden_size)
                gtf = gemm_flops / (gemm_ms * 1e-3) / 1e12
                res["roofline_gemm"] = roofline_mfma("gemm_kernel_v8 (all instantiations: q|k|v, out-proj, fc1, fc2, patch embedding)", gtf, iu["gemm"] if iu else None,
                                                     {"traffic": None, "ms_per_step": gemm_ms})
            if fam_steps != args.steps:
                res["kernel_family_note"] = "attn: HIP events inside the timed region; gemm / rowops / vlcabs: two extra un-timed steps"
        if world == 1 and not args.no_other_configs:
            # the other single-GPU BASELINE configs + the 1e-3-compliant fp32 mode, a few steps each (not bench lines of
            # their own: the driver times the whole process; parity for these shapes is in tests/test_gpu_fullsize.py)
            del pixels, out
            model.close()
            torch.cuda.empty_cache()
            res["other_configs"] = [
                short_run(sd, cfg, device, "bf16", 16, 1024, 64, "upsample", 8, 32),
                short_run(sd, cfg, device, "f16", 1, 1536, 193, "none", 6, 16),
                short_run(sd, cfg, device, "f16", 32, 1024, 14, "none", 6, 10),      # the <= 5e-3 16-bit mode on the headline shape
                short_run(sd, cfg, device, "f32", 32, 1024, 14, "none", 6, 10),      # the 1e-3 mode, default options ("f32_precision high")
                # the headline shape with its input pipeline inside the step (never `value`): PCIe-inclusive, and from raw detector images
                short_run(sd, cfg, device, "bf16", 32, 1024, 14, "none", 6, 10, pipeline="host"),
                short_run(sd, cfg, device, "bf16", 32, 1024, 14, "none", 6, 10, pipeline="raw"),
                short_run(sd, cfg, device, "bf16", 32, 1024, 14, "none", 6, 10, pipeline="rawhost"),
                short_run(sd, cfg, device, "bf16", 64, 518, 14, "none", 6, 10, steps=10, warmup=3),      # the released model's own resolution and eval batch
                short_run(sd, cfg, device, "f32", 32, 1024, 14, "none", 6, 10, f32_precision="fast"),      # the 1e-3 mode's opt-in level: P V on the f16 hi planes alone
                # round 6: the 1e-3 mode with its input pipeline inside the step — possible since the fp32 forward no longer synchronises the stream (predicated guard)
                short_run(sd, cfg, device, "f32", 32, 1024, 14, "none", 6, 10, pipeline="rawhost"),
                # the 1e-3 mode on the other two single-GPU BASELINE shapes (VERDICT r5 missing #6): configs[3] with per-pixel maps, configs[4]'s per-GPU shape
                short_run(sd, cfg, device, "f32", 16, 1024, 64, "upsample", 8, 32, steps=3, warmup=1),
                short_run(sd, cfg, device, "f32", 1, 1536, 193, "none", 6, 16),
            ]
            res["per_request"] = request_leg(sd, cfg, device)
            res["per_request_518"] = request_leg(sd, cfg, device, S=518)      # the released model's own resolution (radzero.yaml:19): the README's single-image call
            # the same two requests in the mode that meets north_star's 1e-3 (the reference's own inference precision: run.py:135-137, inference/utils.py:37)
            res["batch_driver_518"] = driver_leg(sd, cfg, device)      # 518^2 x 64 through the batch driver, with / without batch shaping
            res["per_request_f32"] = request_leg(sd, cfg, device, dtype="f32", steps=12, warmup=4)
            res["per_request_518_f32"] = request_leg(sd, cfg, device, dtype="f32", S=518, steps=12, warmup=4)
            # north_star's tolerance (1e-3 on logits and maps) is met by the fp32 mode only: its throughput on the SAME shape,
            # stated next to `value` (which is BASELINE configs[1]'s own dtype, bf16)
            f32_leg, fast_leg = res["other_configs"][3], res["other_configs"][8]
            res["value_1e3_mode"] = {"images_per_s": f32_leg["images_per_s"], "steps": f32_leg["steps"], "warmup": f32_leg["warmup"],
                                     "dtype": "f32, default options (f32_precision high): operands as f16 hi planes + correction planes — block-scaled e4m3 MFMAs for the GEMMs' and "
                                              "the attention's P V correction terms, f16 lo planes for the scores; fp32 accumulate",
                                     "f32_split_guard_reruns": f32_leg.get("f32_split_guard_reruns"),
                                     "ms_per_step": f32_leg["ms_per_step"], "roofline": f32_leg.get("roofline"),
                                     "frac_of_mfma_peak_whole_path": f32_leg.get("frac_of_mfma_peak_whole_path"),
                                     "frac_of_issue_bound_whole_path": f32_leg.get("frac_of_issue_bound_whole_path"),
                                     "with_input_pipeline_rawhost_images_per_s": res["other_configs"][9]["images_per_s"],
                                     "error_vs_reference": "not measured by this run: tests/test_gpu_fullsize.py::test_cfg2_full_batch_fp32_default_options gates THIS configuration "
                                                           "(B = 32, default options, the reference's 1024^2 golden inside the batch) and tests/test_gpu_model.py every golden at <= 1e-3 "
                                                           "(FP32_TOL); last recorded maxima 7.5e-5 scores / 2.7e-5 logits on the goldens; on the outlier-channel checkpoint 4.7e-4 at N = 257 (G8) and, round 6, at the timed shapes themselves "
                                                           "2.3e-4 (G14: 1024^2 inside B = 32) / 3.2e-4 (G15: 518^2 inside B = 64) in the MX form, 1.5e-5 / 2.2e-5 alone in the three-plane form "
                                                           "(tests/test_gpu_outlier_timed_shapes.py; profiles/r06/outlier_timed_shapes.log)",
                                     "opt_in_fast": {"images_per_s": fast_leg["images_per_s"], "how": "model.set_f32_precision('fast') / option attn_f32_pv = 1",
                                                     "error_vs_reference": "3.1e-4 scores / 8.9e-5 logits worst over the goldens but, on the outlier-channel checkpoint, 1.7e-3 at 224^2 (G8) and 518^2 (G15) and 7.1e-4 at 1024^2 (G14) — "
                                                                           "outside the 1e-3 contract on two of the three, hence opt-in (profiles/r06/f32_fast_on_outlier_fixtures.txt, profiles/r05/fp32_term_ablation.log)"}}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg, sd, S, T, ids, mask, eager=args.cpu_eager, all_cores=args.cpu_all_cores)
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
