#!/usr/bin/env python3
"""Benchmark of the fused STFT -> dB -> u8 hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" is one pass of the hot path over the whole workload: a 1 h, 384 kHz
mono synthetic FM-chirp recording, FFT 1024, Hann, 75 % overlap, u8 dB frames
plus per-frame peak bin / peak magnitude / global max (5 399 997 frames, one
kernel launch), with the samples already resident in HBM.  With --gpus N every
rank holds one such recording of its own (independent recordings, no data-path
collective: weak scaling); only a per-rank summary row is gathered at the end.

Prints ONE JSON line on rank 0.  `roofline` prices the fused kernel against the
HBM peak using the algorithmic bytes per frame (each sample read once, each
output byte written once); `valu` prices the same launch against the fp32 vector
peak, since at 25 flop/byte this kernel sits on the compute side of the ridge.
`cpu_baseline` times the CPU oracle (a C port of the reference's algorithm; the
reference itself is a wasm binary that cannot travel to the GPU box) on a
bounded prefix of the same recording on the host's cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_FFT = 1024
NOVERLAP = 768
HOP = N_FFT - NOVERLAP
SAMPLE_RATE = 384000
SECONDS = 3600
GAIN_DB, RANGE_DB = 20.0, 80.0
HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak
VALU_PEAK_TFLOPS = 157.3        # MI355X fp32 vector peak (FMA = 2 flop)
BYTES_PER_FRAME = 4 * HOP + N_FFT // 2 + 6   # f32 samples in, u8 bins + {u16 bin, f32 mag} out
# complex-FFT convention 5 N log2 N, plus window N, magnitude 4 B, dB + quantise 12 B (B = N/2)
FLOPS_PER_FRAME = 5 * N_FFT * 10 + N_FFT + 16 * (N_FFT // 2)


def cpu_baseline(x, threads, min_seconds=10.0):
    """Times the CPU oracle on `x` (a host prefix of the workload), one engine per
    thread, chunks with an N - hop halo so that chunked frames equal unchunked ones.
    The prefix is processed again and again until at least `min_seconds` have passed."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import sono_oracle as so

    so.build()
    so.lib()
    n = len(x)
    seconds_of_audio = n / SAMPLE_RATE
    total_frames = so.num_frames(n, N_FFT, NOVERLAP)
    per = (total_frames + threads - 1) // threads
    jobs = []
    for t in range(threads):
        f0, f1 = t * per, min((t + 1) * per, total_frames)
        if f0 < f1:
            jobs.append(x[f0 * HOP:(f1 - 1) * HOP + N_FFT])

    def work(chunk):
        e = so.OracleEngine(N_FFT, "hann")
        u8 = e.compute_spectrogram_u8(chunk, NOVERLAP, GAIN_DB, RANGE_DB)
        e.get_peak_magnitudes(0.0)
        return len(u8) // (N_FFT // 2)

    with ThreadPoolExecutor(max_workers=threads) as pool:
        list(pool.map(work, [j[:N_FFT + HOP] for j in jobs]))        # warm-up (library load, page-in)
        done, passes = 0, 0
        t0 = time.perf_counter()
        while True:
            done += sum(pool.map(work, jobs))
            passes += 1
            dt = time.perf_counter() - t0
            if dt >= min_seconds or passes >= 50:
                break
    return {"value": done / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"first {seconds_of_audio:g} s of the same recording ({total_frames} frames) x {passes} passes "
                      f"through oracle/sono_oracle.c, {threads} threads, {dt:.1f} s wall",
            "audio_s_per_s": done * HOP / SAMPLE_RATE / dt}


def measured_traffic():
    """HBM bytes per launch of the timed kernel from the committed rocprofv3 counter passes
    (profiles/hbm_traffic.json, written by tools/profile_summary.py), or None."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if t.get("fft_size") == N_FFT and t.get("noverlap") == NOVERLAP and t.get("seconds") == SECONDS:
            return t
    except (OSError, ValueError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--seconds", type=int, default=SECONDS, help="length of the recording (default: 1 h)")
    ap.add_argument("--cpu-seconds", type=float, default=600.0,
                    help="seconds of audio in the CPU baseline's sample, processed repeatedly for >= 10 s (0 disables it)")
    args = ap.parse_args()

    import torch
    import spectrogram_amd as sa
    from spectrogram_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if distributed:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.cuda.current_device()

    n = args.seconds * SAMPLE_RATE
    frames = sa.frames_for(n, N_FFT, NOVERLAP)
    bins = N_FFT // 2
    x = synth.chirp_torch(0, n, SAMPLE_RATE, stream=rank, device="cuda")
    out = torch.empty(frames * bins, dtype=torch.uint8, device="cuda")
    peak_bin = torch.empty(frames, dtype=torch.int16, device="cuda")
    peak_mag = torch.empty(frames, dtype=torch.float32, device="cuda")
    gmax = torch.zeros(1, dtype=torch.int32, device="cuda")
    engine = sa.SpectrogramEngine(N_FFT, "hann", None, device=dev)
    stream = torch.cuda.Stream()          # the stream every launch of the timed region goes to

    def step():
        with torch.cuda.stream(stream):
            gmax.zero_()
        engine.stft_u8_dev(x.data_ptr(), 1, n, n, NOVERLAP, GAIN_DB, RANGE_DB, out.data_ptr(),
                           peak_bin.data_ptr(), peak_mag.data_ptr(), gmax.data_ptr(), stream.cuda_stream)

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    barrier()
    wall = time.perf_counter() - t0
    gpu_ms = ev0.elapsed_time(ev1)          # HIP events on the launch stream: the kernels' own time

    tmax = torch.tensor([wall], dtype=torch.float64, device="cuda")
    if distributed:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall_max = float(tmax.item())

    # per-rank summary row, gathered with RCCL (the only collective of the path)
    active = int((peak_mag > 0.4 * gmax.view(torch.float32)).sum().item())
    row = torch.tensor([float(rank), float(frames), float(gmax.view(torch.float32).item()), float(active)],
                       dtype=torch.float64, device="cuda")
    rows = [row]
    if distributed:
        rows = [torch.empty_like(row) for _ in range(world)]
        dist.all_gather(rows, row)

    if rank == 0:
        total_frames = frames * world
        ms_per_step = wall_max * 1e3 / args.steps
        kernel_ms = gpu_ms / args.steps
        achieved = frames * BYTES_PER_FRAME / (kernel_ms * 1e-3) / 1e9
        tflops = frames * FLOPS_PER_FRAME / (kernel_ms * 1e-3) / 1e12
        result = {
            "metric": "STFT frames/sec (audio-sec/sec) at FFT 1024, 75% overlap, 384 kHz; % HBM roofline",
            "value": total_frames * args.steps / wall_max,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "audio_s_per_s": total_frames * HOP / SAMPLE_RATE * args.steps / wall_max,
            "config": {
                "workload": f"{args.seconds} s synthetic {SAMPLE_RATE} Hz mono FM-chirp recording per GPU, FFT {N_FFT} Hann, "
                            f"75% overlap (hop {HOP}), uint8 dB frames + per-frame peak bin/magnitude + global max, "
                            f"{frames} frames per GPU, samples resident in HBM",
                "fft_size": N_FFT, "noverlap": NOVERLAP, "sample_rate": SAMPLE_RATE,
                "frames_per_gpu": frames, "parallelism": f"{world} independent recording(s), one per GPU",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "kernel_ms": kernel_ms, "bytes_per_frame": BYTES_PER_FRAME,
                "algorithmic_bytes": frames * BYTES_PER_FRAME,
            },
            "valu": {"achieved": tflops, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / VALU_PEAK_TFLOPS,
                     "flops_per_frame": FLOPS_PER_FRAME},
            "summary_rows": [[float(v) for v in r.tolist()] for r in rows],
        }
        traffic = measured_traffic() if args.seconds == SECONDS else None
        if traffic:
            result["roofline"]["traffic"] = traffic["hbm_bytes"]
            result["roofline"]["traffic_source"] = traffic["source"]
        if world == 1 and args.cpu_seconds > 0:
            threads = min(16, os.cpu_count() or 1)
            n_cpu = int(min(args.cpu_seconds, args.seconds) * SAMPLE_RATE)
            result["cpu_baseline"] = cpu_baseline(x[:n_cpu].cpu().numpy(), threads)
        print(json.dumps(result), flush=True)

    engine.free()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
