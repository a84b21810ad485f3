#!/usr/bin/env python3
"""W4A16 GEMM benchmark on MI355X (BASELINE.json: "W4A16 GEMM TOPS vs M, K=N=4096, g=128").

    python bench.py [--gpus N] [--steps K] [--warmup W]

A step = one W4A16 GEMM launch (through the C ABI) at M=512, K=N=4096, group 128 on synthetic data already
resident in HBM; consecutive steps cycle through enough distinct weight sets to exceed the 256 MiB Infinity
Cache, so small-M numbers are HBM numbers, not cache numbers.  Rank 0 prints ONE JSON line:

  value / ms_per_step  K steps replayed back to back (one hipGraph), bracketed by barrier + synchronize, max over
                       ranks; includes the launch boundary between consecutive kernels
  roofline             the GEMM kernel's OWN duration (hipEvent pair bound to each dispatch, the same clock
                       rocprofv3 --kernel-trace reports) against the HBM or MFMA peak
  sweep                the same two measurements for every M of the BASELINE sweep (1, 8, 64, 512)
  decode_layers        kernel duration and HBM fraction of the Llama-2-7B layer shapes at M=1 (N=1 only)
  prefill_layers       kernel duration and MFMA fraction of prefill-sized launches (N=1 only)
  decode               decode / prefill tok/s of synthetic Llama-2-7B (bs = 1, 64), Mistral-7B (bs = 64) and Llama-2-70B (bs = 16)
                       stacks (128/128, hipGraph step; N=1 only)
  cpu_baseline         the reference's CPU path (dequantize_gemm + torch.matmul, restated in oracle/cpu_path.py)
                       timed on the host cores on a bounded sample, N=1 only

The path does not shard (one dense per-layer GEMM): --gpus N runs N independent replicas.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E, spec (MI355X_MICROARCH.md)
MFMA_PEAK_TFLOPS = 2500.0  # dense f16/bf16 MFMA (MI355X_MICROARCH.md; sparse marketing figure NOT used)


def log(*a):
    print(*a, file=sys.stderr, flush=True)



def algorithmic_bytes(M, K, N, G):
    """int4 weights + fp16 scales + int4 zero points (both un-duplicated) + x + y."""
    return K * N // 2 + (K // G) * N * 2 + (K // G) * N // 2 + 2 * M * K + 2 * M * N


def algorithmic_flops(M, K, N):
    return 2 * M * K * N


def synthetic_layer(M, K, N, G, seed=0):
    """SURVEY.md 8(d) synthetic inputs: integer weights and zero points ~ U{0..15}, scales ~ U(0.005, 0.025) fp16, x ~ N(0, 1)
    fp16 -> (x [M, K], iw [K, N] uint8, s [K/G, N] fp16, z [K/G, N] uint8)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    iw = rng.integers(0, 16, size=(K, N), dtype=np.uint8)
    z = rng.integers(0, 16, size=(K // G, N), dtype=np.uint8)
    s = rng.uniform(0.005, 0.025, size=(K // G, N)).astype(np.float16)
    x = rng.standard_normal((M, K)).astype(np.float16)
    return x, iw, s, z


def pmc_traffic(M, K, N, G, kernel, plan):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 counter pass for this shape
    (profiles/rNN_pmc_m<M>_*.txt, written by tools/prof_passes.sh on K=N=4096, g=128): counters need their own rocprofv3
    run (gpurun refuses --pmc next to anything but --kernel-trace), so this is read back rather than collected inside the
    timed run.  Bytes = 2 * FETCH_SIZE KiB (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md 'HBM') +
    WRITE_SIZE KiB.  The file must be ABOUT the kernel the planner picks today: its '== <kernel>' header is matched against
    the plan's family word, and its SHA-256 goes into the JSON; a stale or foreign file yields traffic = null, not a number."""
    import glob
    import hashlib
    if (G, kernel) != (128, 0):
        return None, None
    here = os.path.dirname(os.path.abspath(__file__))
    # K = N = 4096 (the BASELINE sweep): r*_pmc_m<M>_<family>.txt; other layer shapes [r06]: r*_pmc_<M>x<K>x<N>_<family>.txt
    files = sorted(glob.glob(os.path.join(here, "profiles", f"r*_pmc_m{M}_*.txt"))) if (K, N) == (4096, 4096) else []
    files = sorted(files + glob.glob(os.path.join(here, "profiles", f"r*_pmc_{M}x{K}x{N}_*.txt")), key=os.path.basename)
    if not files:
        return None, None
    text = open(files[-1]).read()
    family = plan.split()[0]                                   # "skinny" | "tiled" | "wide" | "xk"
    heads = [l for l in text.splitlines() if l.startswith("== ")]
    if not heads or not any(f"w4a16_{family}" in h or (family == "wide" and "w4a16_ring" in h) or (family == "skinny" and "w4a16_frag8" in h) for h in heads):
        return None, {"file": "profiles/" + os.path.basename(files[-1]), "rejected": f"profiled kernel is not the planner's ({family})"}
    vals = {}
    for line in text.splitlines():
        f = line.split()
        if len(f) >= 2 and f[0] in ("FETCH_SIZE", "WRITE_SIZE") and f[0] not in vals:
            vals[f[0]] = float(f[1])
    if "FETCH_SIZE" not in vals:
        return None, None
    src = {"file": "profiles/" + os.path.basename(files[-1]), "sha256": hashlib.sha256(text.encode()).hexdigest()[:16],
           "kernel": heads[0][3:].split(":")[0]}
    return (2.0 * vals["FETCH_SIZE"] + vals.get("WRITE_SIZE", 0.0)) * 1024.0, src


def pmc_issue_mix(M, K, N, G, kernel, src):
    """north_star: "evidenced by rocprof MFMA-busy %".  From the same committed counter pass as `traffic` (accepted only if it is
    about the kernel the planner picks today): the share of the launch during which the matrix pipes were busy --
    SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the 1024 SIMDs) / (1024 * SQ_BUSY_CYCLES / 32 shader engines) -- and the VALU-class
    instructions issued per MFMA (SQ_INSTS_VALU counts the MFMAs themselves).  [r05] The denominator was GRBM_GUI_ACTIVE / 8 until
    r04: that counter carries ~20 k cycles of dispatch overhead per launch (30.8 k "active" cycles for a 5.9 us dispatch would be a
    5.2 GHz clock), which under-read every short launch; SQ_BUSY_CYCLES / 32 agrees with the kernels' own s_memtime clocks."""
    if not src or "file" not in src or "rejected" in src:
        return {}
    here = os.path.dirname(os.path.abspath(__file__))
    vals = {}
    for line in open(os.path.join(here, src["file"])):
        f = line.split()
        if len(f) >= 2 and f[0] in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_WAVE_CYCLES",
                                    "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY") and f[0] not in vals:
            vals[f[0]] = float(f[1])
    out = {}
    if vals.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
        out["mfma_busy_frac"] = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * vals["SQ_BUSY_CYCLES"] / 32.0)
        out["mfma_busy_denominator"] = "1024 SIMDs x SQ_BUSY_CYCLES / 32"
    if vals.get("SQ_INSTS_MFMA") and "SQ_INSTS_VALU" in vals:
        out["valu_per_mfma"] = vals["SQ_INSTS_VALU"] / vals["SQ_INSTS_MFMA"]
    if vals.get("SQ_WAVE_CYCLES"):
        out["wave_cycles_waiting_frac"] = vals.get("SQ_WAIT_ANY", 0.0) / vals["SQ_WAVE_CYCLES"]
        out["wave_cycles_issue_stalled_frac"] = vals.get("SQ_WAIT_INST_ANY", 0.0) / vals["SQ_WAVE_CYCLES"]
    return out


def decode_leg(dev, seconds, log):
    """Second half of the BASELINE metric: decode tok/s of a synthetic Llama-2-7B AWQ-QUICK stack at bs = 1 and 64,
    prefill/decode = 128/128, the reference's methodology (examples/benchmark.py:38-67,127-129: tok/s = bs / median step),
    one decode step captured in a hipGraph.  Runs after the timed GEMM steps; `seconds` bounds it (fewer decode steps)."""
    import time

    import numpy as np
    import torch
    from quick_amd.decoder import CONFIGS, SyntheticDecoder, run_generation
    out = []
    t0 = time.perf_counter()
    # BASELINE.json configs[2..4]: Llama-2-7B bs = 1 (and 64, the metric's second batch size), Mistral-7B bs = 64, Llama-2-70B bs = 16
    for name, bs in (("llama2-7b", 1), ("llama2-7b", 64), ("mistral-7b", 64), ("llama2-70b", 16)):
        if time.perf_counter() - t0 > seconds:
            out.append({"model": name, "batch": bs, "skipped": "decode-seconds budget spent"})
            continue
        cfg = CONFIGS[name]
        model = SyntheticDecoder(cfg, bs, 256, dev)
        torch.cuda.synchronize()
        run_generation(model, 128, 8, use_graph=False, fused=True)                  # warm-up (allocator, lazy init)
        gen = 128 if time.perf_counter() - t0 < 0.6 * seconds else 32
        prefill, steps = run_generation(model, 128, gen, use_graph=True, fused=True)
        med = float(np.median(steps))
        out.append({"model": cfg.name, "batch": bs, "prefill_len": 128, "decode_len": gen, "tok_s": bs / med, "ms_per_step": med * 1e3,
                    "prefill_tok_s": 128 * bs / prefill, "weight_stream_GBs": model.weight_bytes() / med / 1e9,
                    "launch": "hipgraph", "data": "synthetic random weights"})
        log(f"decode {cfg.name} bs={bs}: {bs / med:9.1f} tok/s  {med * 1e3:.3f} ms/step  weights at {model.weight_bytes() / med / 1e9:.0f} GB/s")
        del model
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)   # one graph replay costs ~0.4 ms on top of its K launches: 7 % at K = 200
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--timed-launch", default="auto", choices=("auto", "graph", "eager"),
                    help="how the K timed steps reach the GPU: one hipGraph replay, or launches queued behind the untimed warm-up (auto: eager up to 256 steps)")
    ap.add_argument("--M", type=int, default=512, help="token count of the headline step")
    ap.add_argument("--K", type=int, default=4096)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--G", type=int, default=128)
    ap.add_argument("--sweep", default="1,8,64,512", help="comma-separated M values reported in 'sweep'")
    ap.add_argument("--sets", type=int, default=0, help="distinct weight sets cycled through (0 = enough for > 320 MiB)")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 skinny, 2 tiled")
    ap.add_argument("--split-k", type=int, default=0, help="K slices across workgroups (0 = the planner's choice; tuning only)")
    ap.add_argument("--layers", default="1x4096x12288,1x4096x22016,1x11008x4096,16x8192x57344",
                    help="MxKxN shapes (Llama-2-7B fused qkv, gate_up, down at bs=1; Llama-2-70B gate_up at bs=16) timed kernel-only into 'decode_layers'; '' = none")
    ap.add_argument("--prefill-layers", default="4096x4096x4096,8192x4096x22016,8192x11008x4096,4096x28672x8192",
                    help="MxKxN prefill-sized launches (K = N = 4096; Llama-2-7B gate_up / down at 8192 tokens; Llama-2-70B down) into 'prefill_layers'; '' = none")
    ap.add_argument("--cpu-seconds", type=float, default=14.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--decode-seconds", type=float, default=240.0,
                    help="budget of the decode tok/s leg (Llama-2-7B bs=1,64; Mistral-7B bs=64; Llama-2-70B bs=16; 0 = skip)")
    args = ap.parse_args()

    import numpy as np
    import torch

    from quick_amd import replicas
    rank, local_rank, world = replicas.world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the W4A16 GEMM has no CPU implementation")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = replicas.init("gloo")             # replicas only, no RCCL anywhere: the barrier and the max over ranks run on CPU tensors

    from quick_amd import _lib, kernels, packing
    from quick_amd.build import build
    build()
    lib = _lib.load()

    K, N, G = args.K, args.N, args.G
    Ms = sorted({int(m) for m in args.sweep.split(",") if m} | {args.M})
    set_bytes = K * N // 2 + (K // G) * 2 * N * 2 + (K // G) * (N // 4) * 4
    n_sets = args.sets or max(2, -(-(320 << 20) // set_bytes))

    # ---- synthetic data: set 0 from logical tensors (shared with the CPU baseline), the rest raw bits.
    # (Made here, not by oracle/: the oracle is the checker of the cpu_baseline leg below and of the tests, nothing the GPU legs use.)
    x_np, iw, s, z = synthetic_layer(max(Ms), K, N, G, seed=0)
    x_full = torch.from_numpy(x_np).to(dev)
    sets = [tuple(t.contiguous() for t in packing.pack_mi355x(torch.from_numpy(iw).to(dev), torch.from_numpy(s).to(dev),
                                                              torch.from_numpy(z.astype(np.int32)).to(dev)))]
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    for _ in range(n_sets - 1):
        qw, sc, qz = packing.random_mi355x(K, N, G, dev, gen)
        sets.append((qw, sc, qz))
    arr = lambda i: (ctypes.c_void_p * n_sets)(*[st[i].data_ptr() for st in sets])
    qw_arr, sc_arr, qz_arr = arr(0), arr(1), arr(2)
    stream = torch.cuda.current_stream()

    flush_buf = torch.empty(512 << 20, dtype=torch.uint8, device=dev)

    def flush_cache():
        flush_buf.fill_(1)
        flush_buf.view(torch.int64).sum()
        torch.cuda.synchronize()

    def measure(M, steps, warmup):
        x = x_full[:M].contiguous()
        y = torch.empty((M, N), dtype=torch.float16, device=dev)
        ws_bytes = lib.quick_w4a16_workspace_bytes_ex(M, K, N, G, args.kernel, args.split_k)
        ws = torch.zeros(max(ws_bytes, 1), dtype=torch.uint8, device=dev)   # zero-filled once; the library keeps it so

        def launch(i):
            qw, sc, qz = sets[i % n_sets]
            rc = lib.quick_w4a16_gemm_f16_ex(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), None, y.data_ptr(),
                                             ws.data_ptr(), ws_bytes, M, K, N, G, args.kernel, args.split_k,
                                             torch.cuda.current_stream().cuda_stream)
            if rc != 0:
                raise RuntimeError(_lib.last_error())

        for i in range(warmup):
            launch(i)
        torch.cuda.synchronize()
        # K steps as ONE hipGraph: back-to-back on the GPU, no host launch cost in the timed region
        mode = "hipgraph"
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph):
                for i in range(steps):
                    launch(warmup + i)
            graph.replay()                       # untimed first replay (graph upload)
        except Exception as e:                   # pragma: no cover
            log(f"graph capture failed ({e}); timing eager launches")
            graph, mode = None, "eager"
        torch.cuda.synchronize()
        # The untimed replay touched exactly the weight sets the timed one will: with few steps (the driver's --steps 20 = 181 MB)
        # they would all still sit in the 256 MiB Infinity Cache.  Push them out: write and read 512 MiB of something else.
        flush_cache()
        # ... and bring the clocks back up on weight sets the timed replay does not touch (the flush is memory-bound: the first
        # launches behind it would otherwise read the clock ramp, 20 steps are only 0.1-0.5 ms)
        spare = [j % n_sets for j in range(warmup + steps, warmup + steps + n_sets) if (j % n_sets) not in {(warmup + i) % n_sets for i in range(steps)}]
        # [r05] 12 launches were 0.05-0.26 ms: shorter than the clock ramp (the driver's 20-step run read the M = 512 step 1.4 us above the
        # kernel's own duration measured later in the same process).  150 launches round-robin over the spare sets = 0.6-3.5 ms.
        # [r05] ... and NO host synchronisation between them and the timed steps: barrier + synchronize come first, then the untimed launches,
        # the first event, the K graph-replayed steps and the second event go into the stream back to back.  With a synchronize in front of
        # the first event the GPU sat idle while the host submitted the graph: the 20-step protocol read M = 512 at 24.3 us per step against
        # 20.5 with 2000 steps in the same process (M = 1: 4.5 against 4.0) -- the idle gap and the clock ramp behind it, not the kernels.
        # [r05] ... and the untimed launches are 40 replays of a 50-launch hipGraph on the spare sets (2000 launches, 8-45 ms): 150 launches
        # (0.6-3.5 ms) still left the 20-step M = 512 step at 24.2 us against 21.9 behind 2000 (gpu_m512.sh, one box, one session).
        # [r05] ... and with few steps the timed launches are queued EAGERLY behind that untimed work instead of replayed as a graph: the
        # host gets K <= 256 launches into the queue long before the GPU reaches them, so they run back to back exactly as graph nodes do,
        # without the ~30-50 us a graph replay needs to start (M = 512, 20 steps: graph 21.9 us per step, queued eager 21.2-21.5, 2000-step
        # graph 21.2; M = 1: 4.55 / 4.13 / 3.94).  Above 256 steps the graph is kept: its start is amortised and the host could fall behind.
        n_warm, gwarm = (2000 if spare else 0), None
        if spare and graph is not None:
            try:
                gwarm = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gwarm):
                    for r in range(50):
                        launch(spare[r % len(spare)])
            except Exception:                    # pragma: no cover
                gwarm = None
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        if gwarm is not None:
            for _ in range(n_warm // 50):
                gwarm.replay()
        else:
            for r in range(n_warm):
                launch(spare[r % len(spare)])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        queued_eager = args.timed_launch == "eager" or (args.timed_launch == "auto" and steps <= 256 and n_warm > 0)
        if graph is not None and not queued_eager:
            graph.replay()
        else:
            for i in range(steps):
                launch(warmup + i)
        e1.record()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        ms_step = replicas.max_over_ranks(dist, e0.elapsed_time(e1) / steps)
        if graph is not None and queued_eager:
            mode = "queued-eager"

        # the kernel's own duration: event pair bound to each dispatch, cycling the same weight sets
        # per-dispatch event pairs: 200 launches whatever --steps says (the mean of 15 separated launches wanders by +-8 % from run
        # to run -- clocks -- while rocprofv3's average over thousands does not; 200 agree with it within ~2 %)
        psteps = min(max(steps, 200), 300)
        kus = (ctypes.c_float * psteps)()
        rc = lib.quick_w4a16_gemm_profile(x.data_ptr(), qw_arr, sc_arr, qz_arr, n_sets, y.data_ptr(), ws.data_ptr(), ws_bytes,
                                          M, K, N, G, args.kernel, args.split_k, psteps, kus, stream.cuda_stream)
        if rc != 0:
            raise RuntimeError(_lib.last_error())
        k_us_events = float(np.mean(np.asarray(kus[:])[min(5, psteps - 1):]))
        # The dispatch-bound event pair over-reads on some boxes (r01: 0.7 % above the step; r02: 27.1 us against a 24.4 us step
        # and rocprofv3's 25.1 us average in the same session).  A launch cannot take longer than the back-to-back step of
        # identical launches it is part of, so the step bounds it; both numbers go into the JSON.
        k_us = min(k_us_events, ms_step * 1e3) if mode in ("hipgraph", "queued-eager") else k_us_events
        # same, cache-resident (one weight set): what a launch sees when the layer was just touched
        rc = lib.quick_w4a16_gemm_profile(x.data_ptr(), qw_arr, sc_arr, qz_arr, 1, y.data_ptr(), ws.data_ptr(), ws_bytes,
                                          M, K, N, G, args.kernel, args.split_k, psteps, kus, stream.cuda_stream)
        k_us_hot = float(np.mean(np.asarray(kus[:])[min(5, psteps - 1):])) if rc == 0 else None

        # in-kernel span (first wave's start -> last wave's end, 100 MHz counter stamped by the kernel itself): the clock
        # that can see a launch shorter than the ~4.2 us an EMPTY kernel reads on the dispatch-duration clock
        nspan = min(psteps, 48)
        sus = (ctypes.c_float * nspan)()
        rc = lib.quick_w4a16_gemm_span(x.data_ptr(), qw_arr, sc_arr, qz_arr, n_sets, y.data_ptr(), ws.data_ptr(), ws_bytes,
                                       M, K, N, G, args.kernel, args.split_k, nspan, sus, stream.cuda_stream)
        k_us_span = float(np.median(np.asarray(sus[:])[min(5, nspan - 1):])) if rc == 0 else None

        flops, nbytes = algorithmic_flops(M, K, N), algorithmic_bytes(M, K, N, G)
        ridge = MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
        if flops / nbytes < ridge:
            roof = {"bound": "hbm", "achieved": nbytes / (k_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
        else:
            roof = {"bound": "mfma", "achieved": flops / (k_us * 1e-6) / 1e12, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s"}
        roof["frac"] = roof["achieved"] / roof["peak"]
        if k_us_span:
            work = nbytes / 1e9 if roof["bound"] == "hbm" else flops / 1e12
            roof["kernel_us_inkernel"] = k_us_span
            roof["achieved_inkernel"] = work / (k_us_span * 1e-6)
            roof["frac_inkernel"] = roof["achieved_inkernel"] / roof["peak"]
        pbuf = ctypes.create_string_buffer(256)
        lib.quick_w4a16_plan_describe(M, K, N, G, args.kernel, args.split_k, pbuf, 256)
        roof["plan"] = pbuf.value.decode()
        roof["traffic"], roof["traffic_source"] = pmc_traffic(M, K, N, G, args.kernel, roof["plan"])
        roof["traffic_measured_in_this_run"] = False   # counters need their own rocprofv3 --pmc passes: read back from the committed profiles/ file named in traffic_source
        roof.update({"kernel_us": k_us, "kernel_us_event_pairs": k_us_events, "kernel_us_cache_resident": k_us_hot,
                     "algorithmic_bytes": nbytes, "flops": flops})
        roof.update(pmc_issue_mix(M, K, N, G, args.kernel, roof["traffic_source"]))
        # the ONE headline fraction: whole-step throughput (what `value` is made of) over the peak; `frac` is the same work over the
        # kernel's own dispatch duration (the clock rocprofv3 --kernel-trace reads), `frac_inkernel` over the first-wave-in -> last-wave-out span
        step_work = (nbytes / 1e9 if roof["bound"] == "hbm" else flops / 1e12) / (ms_step * 1e-3)
        roof["frac_step"] = step_work / roof["peak"]
        roof["headline_fraction"] = "frac_step"
        roof["frac_clocks"] = {"frac_step": "back-to-back step (graph replay or queued launches), event to event on the stream", "frac": "dispatch duration (event pair = rocprofv3 kernel trace)",
                               "frac_inkernel": "per-wave s_memrealtime stamps, first wave in -> last wave out"}
        return {"M": M, "ms_per_step": ms_step, "tops": flops / (ms_step * 1e-3) / 1e12, "tops_kernel_only": flops / (k_us * 1e-6) / 1e12,
                "launch": mode, "weight_sets_in_timed_region": min(steps, n_sets), "cache_flushed_before_timed_region": True,
                "clock_warmup_launches_on_other_weight_sets": n_warm,
                "timed_region": "two hipEvents on the stream around exactly K steps (K <= 256: launches queued behind the untimed ones, "
                                "K > 256: one hipGraph replay); barrier + synchronize before the untimed launches that precede the first "
                                "event and after the second event",
                "roofline": roof}, y

    fl = (ctypes.c_float * 60)()
    lib.quick_amd_dispatch_floor(60, fl, stream.cuda_stream)
    floor_us = float(np.median(np.asarray(fl[:])[5:]))
    if rank == 0:
        log(f"dispatch floor (empty 256x512 kernel, same event-pair clock): {floor_us:.2f} us")

    results = {}
    y_head = None
    for M in Ms:
        steps = args.steps
        res, y = measure(M, steps, args.warmup)
        res["roofline"]["empty_kernel_us"] = floor_us
        results[M] = res
        if M == args.M:
            y_head = y.clone()
        if rank == 0:
            r = res["roofline"]
            log(f"M={M:4d}  step {res['ms_per_step'] * 1e3:8.2f} us  kernel {r['kernel_us']:8.2f} us (cache-resident "
                f"{r['kernel_us_cache_resident']:.2f}, in-kernel span {r.get('kernel_us_inkernel', float('nan')):.2f})  {res['tops']:8.2f} TOPS  "
                f"roofline[{r['bound']}] {r['achieved']:.1f} {r['unit']} = {r['frac'] * 100:.1f}%  (in-kernel {r.get('frac_inkernel', float('nan')) * 100:.1f}%)")

    head = results[args.M]
    out = {
        "metric": "w4a16_gemm_tops", "value": replicas.job_throughput(head["roofline"]["flops"], head["ms_per_step"], world) / 1e12, "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"W4A16 GEMM M={args.M} K={K} N={N} group_size={G} (BASELINE.json configs[1])", "M": args.M, "K": K, "N": N,
                   "group_size": G, "weight_sets_cycled": n_sets, "weight_set_bytes": set_bytes, "launch": head["launch"],
                   "weight_sets_in_timed_region": head["weight_sets_in_timed_region"], "cache_flushed_before_timed_region": True,
                   "parallelism": "replicas only" if world > 1 else "single GPU"},
        "roofline": head["roofline"],
        "sweep": [results[m] for m in Ms],
    }

    # ---- decode-layer shapes, kernel duration only (HBM-cold: the weight sets cycled exceed the Infinity Cache)
    if world == 1 and args.layers:
        out["decode_layers"] = []
        for spec in args.layers.split(","):
            Ml, Kl, Nl = (int(v) for v in spec.lower().split("x"))
            sb = Kl * Nl // 2 + (Kl // G) * 2 * Nl * 2 + (Kl // G) * (Nl // 4) * 4
            ns = max(2, -(-(320 << 20) // sb))
            lsets = []
            for _ in range(ns):
                qw, sc, qz = packing.random_mi355x(Kl, Nl, G, dev, gen)
                lsets.append((qw, sc, qz))
            larr = lambda i: (ctypes.c_void_p * ns)(*[st[i].data_ptr() for st in lsets])
            xl = (torch.randn((Ml, Kl), device=dev, generator=gen) * 0.5).half()
            yl = torch.empty((Ml, Nl), dtype=torch.float16, device=dev)
            wsb = lib.quick_w4a16_workspace_bytes_ex(Ml, Kl, Nl, G, args.kernel, 0)
            wsl = torch.zeros(max(wsb, 1), dtype=torch.uint8, device=dev)
            it = 60
            kus = (ctypes.c_float * it)()
            rc = lib.quick_w4a16_gemm_profile(xl.data_ptr(), larr(0), larr(1), larr(2), ns, yl.data_ptr(), wsl.data_ptr(), wsb,
                                              Ml, Kl, Nl, G, args.kernel, 0, it, kus, stream.cuda_stream)
            if rc != 0:
                raise RuntimeError(_lib.last_error())
            k_us = float(np.mean(np.asarray(kus[:])[5:]))
            rc = lib.quick_w4a16_gemm_span(xl.data_ptr(), larr(0), larr(1), larr(2), ns, yl.data_ptr(), wsl.data_ptr(), wsb,
                                           Ml, Kl, Nl, G, args.kernel, 0, 40, kus, stream.cuda_stream)
            s_us = float(np.median(np.asarray(kus[:40])[5:])) if rc == 0 else None
            nb = algorithmic_bytes(Ml, Kl, Nl, G)
            ach = nb / (k_us * 1e-6) / 1e9
            lplan = kernels.plan_describe(Ml, Kl, Nl, G, args.kernel)
            lroof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes": nb,
                     "kernel_us_inkernel": s_us, "frac_inkernel": nb / (s_us * 1e-6) / 1e9 / HBM_PEAK_GBS if s_us else None, "plan": lplan}
            # [r06] HBM traffic of this shape's launch from its committed counter pass (profiles/r*_pmc_<M>x<K>x<N>_*.txt), accepted only if it is about the planner's kernel
            ltraffic, lsrc = pmc_traffic(Ml, Kl, Nl, G, args.kernel, lplan)
            lroof.update({"traffic": ltraffic if lsrc and "rejected" not in lsrc else None, "traffic_source": lsrc, "traffic_measured_in_this_run": False})
            if lsrc and "rejected" not in lsrc:
                lroof.update(pmc_issue_mix(Ml, Kl, Nl, G, args.kernel, lsrc))
            out["decode_layers"].append({"M": Ml, "K": Kl, "N": Nl, "kernel_us": k_us, "weight_sets_cycled": ns, "plan": lplan, "roofline": lroof})
            log(f"layer M={Ml} K={Kl} N={Nl}: kernel {k_us:7.2f} us  {ach:7.1f} GB/s = {100 * ach / HBM_PEAK_GBS:.1f}% of HBM peak"
                + (f"; in-kernel span {s_us:.2f} us = {100 * nb / (s_us * 1e-6) / 1e9 / HBM_PEAK_GBS:.1f}%" if s_us else ""))
            del lsets, larr

    # ---- [r06] what a one-token launch is made of: kernel time = fixed + bytes / rate over the M = 1 rows (the sweep's 4096 x 4096 and the decode layers),
    #      least squares on the dispatch clock and on the in-kernel span
    if world == 1 and out.get("decode_layers"):
        pts = [(r["roofline"]["algorithmic_bytes"], r["kernel_us"], r["roofline"].get("kernel_us_inkernel")) for r in out["decode_layers"] if r["M"] == 1]
        if 1 in results:
            pts.append((results[1]["roofline"]["algorithmic_bytes"], results[1]["roofline"]["kernel_us"], results[1]["roofline"].get("kernel_us_inkernel")))
        if len(pts) >= 3:
            model = {"rows": len(pts), "what": "kernel_us = fixed_us + algorithmic bytes / rate, least squares over the M = 1 rows (sweep + decode_layers)"}
            for name, col in (("dispatch_clock", 1), ("inkernel_span", 2)):
                p2 = [(b, t[col - 1]) for b, *t in pts if t[col - 1]]
                if len(p2) >= 3:
                    A = np.array([[1.0, b] for b, _ in p2])
                    coef, *_ = np.linalg.lstsq(A, np.array([t for _, t in p2]), rcond=None)
                    model[name] = {"fixed_us": float(coef[0]), "marginal_TBps": float(1e-6 / coef[1]) if coef[1] > 0 else None,
                                   "residual_us_max": float(np.abs(A @ coef - np.array([t for _, t in p2])).max())}
            out["small_m_model"] = model
            log(f"small-M model: {model}")

    # ---- prefill-sized launches (compute-bound end of the path), kernel duration only, MFMA roofline
    if world == 1 and args.prefill_layers:
        out["prefill_layers"] = []
        for spec in args.prefill_layers.split(","):
            Ml, Kl, Nl = (int(v) for v in spec.lower().split("x"))
            sb = Kl * Nl // 2 + (Kl // G) * 2 * Nl * 2 + (Kl // G) * (Nl // 4) * 4
            ns = max(2, min(8, -(-(320 << 20) // sb)))
            lsets = [packing.random_mi355x(Kl, Nl, G, dev, gen) for _ in range(ns)]
            larr = lambda i: (ctypes.c_void_p * ns)(*[st[i].data_ptr() for st in lsets])
            xl = (torch.randn((Ml, Kl), device=dev, generator=gen) * 0.5).half()
            yl = torch.empty((Ml, Nl), dtype=torch.float16, device=dev)
            wsb = lib.quick_w4a16_workspace_bytes_ex(Ml, Kl, Nl, G, args.kernel, 0)
            wsl = torch.zeros(max(wsb, 1), dtype=torch.uint8, device=dev)
            it = 32   # three rounds, the best median counts (same as tools/wide_probe.py): the clocks ramp for tens of
            kus = (ctypes.c_float * it)()   # milliseconds after the light launches before this leg
            meds = []
            for _ in range(3):
                rc = lib.quick_w4a16_gemm_profile(xl.data_ptr(), larr(0), larr(1), larr(2), ns, yl.data_ptr(), wsl.data_ptr(), wsb,
                                                  Ml, Kl, Nl, G, args.kernel, 0, it, kus, stream.cuda_stream)
                if rc != 0:
                    raise RuntimeError(_lib.last_error())
                meds.append(float(np.median(np.asarray(kus[:])[4:])))
            k_us = min(meds)
            fl = algorithmic_flops(Ml, Kl, Nl)
            ach = fl / (k_us * 1e-6) / 1e12
            lplan = kernels.plan_describe(Ml, Kl, Nl, G, args.kernel)
            lroof = {"bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS, "flops": fl}
            # (north_star: "evidenced by rocprof MFMA-busy %" -- where a committed counter pass is about this shape and this plan's kernel)
            ltraffic, lsrc = pmc_traffic(Ml, Kl, Nl, G, args.kernel, lplan)
            if lsrc and "rejected" not in lsrc:
                lroof.update({"traffic": ltraffic, "traffic_source": lsrc, "traffic_measured_in_this_run": False})
                lroof.update(pmc_issue_mix(Ml, Kl, Nl, G, args.kernel, lsrc))
            out["prefill_layers"].append({"M": Ml, "K": Kl, "N": Nl, "kernel_us": k_us, "plan": lplan, "kernel_us_medians": meds, "roofline": lroof})
            log(f"prefill M={Ml} K={Kl} N={Nl}: kernel {k_us:8.2f} us  {ach:7.1f} TFLOP/s = {100 * ach / MFMA_PEAK_TFLOPS:.1f}% of the f16 MFMA peak")
            del lsets, larr

    # ---- the reference's CPU path on the host cores, bounded sample, rank 0 / N=1 only
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        from oracle import cpu_path
        # host cores actually available to this process (affinity mask and cgroup quota), not the box's core count
        cores = len(os.sched_getaffinity(0))
        try:
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
            if quota != "max":
                cores = max(1, min(cores, int(int(quota) / int(period))))
        except Exception:
            pass
        torch.set_num_threads(cores)
        # Bounded sample: the reference path on the first n_cpu output channels of the SAME layer and the same x
        # (TOPS is intensive, so a column slice measures the same rate).  The slice is sized from a 128-channel probe
        # so that the leg stays within --cpu-seconds whatever the host's fp16 GEMM speed is.
        def cpu_call(ncols):
            qw_g, qz_g = cpu_path.pack_gemm_format(iw[:, :ncols], z[:, :ncols])
            qw_t, qz_t, s_t = torch.from_numpy(qw_g), torch.from_numpy(qz_g), torch.from_numpy(np.ascontiguousarray(s[:, :ncols]))
            t0 = time.perf_counter()
            y = cpu_path.forward(x_t, qw_t, qz_t, s_t, G)
            return y, time.perf_counter() - t0
        x_t = torch.from_numpy(x_np[:args.M])
        cpu_call(128)                                    # page-in, thread pool
        # grow the slice geometrically while a call stays well inside the budget (host fp16 GEMM speed varies by
        # orders of magnitude between CPUs, and not linearly in the slice width)
        n_cpu, spent = 128, 0.0
        y_cpu, dt1 = cpu_call(n_cpu)
        spent += dt1
        while n_cpu < N and dt1 * 2.5 * 3 < (args.cpu_seconds - spent):     # leave room for >= 3 timed calls at the final width
            n_cpu = min(N, n_cpu * 2)
            y_cpu, dt1 = cpu_call(n_cpu)
            spent += dt1
        times = [dt1]
        while (len(times) < 3 or sum(times) + dt1 < (args.cpu_seconds - spent)) and len(times) < 50:
            y_cpu, dt1 = cpu_call(n_cpu)
            times.append(dt1)
        reps, total = len(times), sum(times)
        dt = float(np.median(times))
        # where the time goes (one more call, split): the per-call dequantisation vs the fp16 matmul -- hosts without a native fp16 GEMM
        # path read hundreds of times slower than BASELINE.md's 73 ms for the full layer (5.8 s per 1024 channels)
        qw_g, qz_g = cpu_path.pack_gemm_format(iw[:, :n_cpu], z[:, :n_cpu])
        t0 = time.perf_counter()
        w_cpu = cpu_path.dequantize_gemm(torch.from_numpy(qw_g), torch.from_numpy(qz_g), torch.from_numpy(np.ascontiguousarray(s[:, :n_cpu])), G)
        t_deq = time.perf_counter() - t0
        t0 = time.perf_counter()
        torch.matmul(x_t, w_cpu)
        t_mm = time.perf_counter() - t0
        # companion figure: the same product in fp32 (hosts whose BLAS has no fast fp16 GEMM read the fp16 line hundreds of times slower than
        # the arithmetic warrants; the reference's CPU path IS the fp16 one, this one only makes the line readable)
        x32, w32 = x_t.float(), w_cpu.float()
        torch.matmul(x32, w32)
        t0 = time.perf_counter()
        torch.matmul(x32, w32)
        t_mm32 = time.perf_counter() - t0
        full_layer_ms = dt * 1e3 * N / n_cpu
        out["cpu_baseline"] = {
            "value": algorithmic_flops(args.M, K, n_cpu) / dt / 1e12, "unit": "TFLOP/s", "cores": cores,
            "kind": "port", "ms_per_call": dt * 1e3, "calls": reps, "ms_per_call_min_max": [min(times) * 1e3, max(times) * 1e3],
            "sample": f"{reps} call(s) of the reference CPU path (dequantize_gemm + torch.matmul, dequant redone per call, "
                      f"oracle/cpu_path.py) on output channels 0..{n_cpu - 1} of the M={args.M} K={K} N={N} g={G} layer, "
                      f"{total:.1f} s on {cores} torch threads",
            "split_ms": {"dequantize": t_deq * 1e3, "matmul_fp16": t_mm * 1e3},
            "companion_fp32": {"kind": "port-fp32", "value": algorithmic_flops(args.M, K, n_cpu) / (t_deq + t_mm32) / 1e12, "unit": "TFLOP/s",
                               "ms_per_call": (t_deq + t_mm32) * 1e3, "what": "the same dequantisation + torch.matmul in fp32 on the same slice, one call"},
            "full_layer_ms_extrapolated": full_layer_ms, "baseline_md_full_layer_ms": 73.3,
            "deviates_over_10x_from_baseline_md": bool(full_layer_ms > 733.0 or full_layer_ms < 7.33),
            "torch_parallel_info": torch.__config__.parallel_info().strip().splitlines()[:6],
        }
        # the GPU result of the headline step's set 0 against the CPU baseline's output (same inputs)
        qw, sc, qz = sets[0]
        from quick_amd import gemm_forward
        y_gpu = gemm_forward(x_full[:args.M].contiguous(), qw, sc, qz, kernel_id=args.kernel).float().cpu()
        out["parity_rel_err_vs_cpu_baseline"] = float((y_gpu[:, :n_cpu] - y_cpu.float()).abs().max() / y_cpu.float().abs().max())

    # ---- decode tok/s (BASELINE metric, second half), after everything that is timed above
    if rank == 0 and world == 1 and args.decode_seconds > 0:
        try:
            out["decode"] = decode_leg(dev, args.decode_seconds, log)
        except Exception as e:                                   # pragma: no cover  (never lose the GEMM line to the extra leg)
            out["decode"] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
