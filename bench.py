#!/usr/bin/env python3
"""Benchmark of the fused STFT -> dB -> u8 hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" is one pass of the hot path over the whole workload: a 1 h, 384 kHz
mono synthetic FM-chirp recording, FFT 1024, Hann, 75 % overlap, u8 dB frames
plus per-frame peak bin / peak magnitude / global max (5 399 997 frames, one
kernel launch), with the samples already resident in HBM.  With --gpus N every
rank holds one such recording of its own (independent recordings, no data-path
collective: weak scaling); only a per-rank summary row is gathered at the end.
Started under torch.distributed.run the script is one rank of WORLD_SIZE (which
must equal N); started plainly with --gpus N > 1 it launches the N ranks itself
(one per GPU, rendezvous on 127.0.0.1) and fails if fewer than N GPUs are visible.

--workload selects another configuration of the same path (cfg3: a batch of 5 s
500 kHz files spread over the ranks, strong scaling; cfg4: FFT 4096 Blackman 95 %
overlap over 10 min; cfg5: survey pass over a long channel per rank -- per-frame
peak features at 75 % overlap plus the detector's fast scan at 50 % overlap, no u8
rows -- cfg5 with the whole detector behind it and the call-summary rows gathered
over the ranks, cfg5_pcm16 up to the fast scan's run list on 16-bit PCM); those
lines carry their own workload description and the same fields.

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


def end_to_end(sa, torch, engine, x_dev, n_fft, noverlap, frames, bins):
    """What a host caller of compute_spectrogram_u8 sees (the wasm glue copies in and out as well,
    spectrogram_wasm.js:58-63, 313-320): the recording in pageable host memory, the rows into a host buffer,
    through sono_stft_u8's chunked pipeline; and beside it the plain PCIe copies of the same bytes from and to
    pinned memory.  Outside the timed region; never part of `value`."""
    import ctypes as C
    import numpy as np
    x = x_dev.cpu().numpy()
    out = np.zeros(frames * bins, dtype=np.uint8)                 # touched: no page faults inside the timing
    got = C.c_size_t(0)
    f32p, u8p = C.POINTER(C.c_float), C.POINTER(C.c_uint8)

    def call():
        t0 = time.perf_counter()
        sa._check(sa.lib().sono_stft_u8(engine._handle(), x.ctypes.data_as(f32p), x.size, noverlap, GAIN_DB, RANGE_DB,
                                        out.ctypes.data_as(u8p), out.size, C.byref(got)))
        return time.perf_counter() - t0
    call()                                                        # first call builds the pinned and device rings
    dt = min(call() for _ in range(3))
    pin_in = torch.empty(x.size, dtype=torch.float32, pin_memory=True)
    pin_out = torch.empty(out.size, dtype=torch.uint8, pin_memory=True)
    dev_out = torch.empty(out.size, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter(); x_dev.copy_(pin_in, non_blocking=True); torch.cuda.synchronize(); h2d = time.perf_counter() - t0
    t0 = time.perf_counter(); pin_out.copy_(dev_out, non_blocking=True); torch.cuda.synchronize(); d2h = time.perf_counter() - t0
    x_dev.copy_(torch.from_numpy(x))                              # the recording back in place
    return {"ms": dt * 1e3, "frames_per_s": got.value / dt, "audio_s_per_s": got.value * (n_fft - noverlap) / SAMPLE_RATE / dt,
            "GBps_h2d": x.nbytes / h2d / 1e9, "GBps_d2h": out.nbytes / d2h / 1e9,
            "pcie_floor_ms": max(h2d, d2h) * 1e3, "over_pcie_floor": dt / max(h2d, d2h),
            "note": "host pointers in and out (pageable memory), sono_stft_u8: chunks of frames through pinned rings, copy-in, upload, "
                    "kernel, download and copy-out overlapped; GBps_* = one plain copy of the same bytes from / to pinned memory"}


def library_sha256(path=None):
    """sha256 of the library build that is being timed (the file the package loads), or None."""
    import hashlib
    path = path or os.environ.get("SONO_LIB_PATH") or os.path.join(ROOT, "spectrogram_amd", "lib", "libsonostft.so")
    try:
        with open(path, "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()
    except OSError:
        return None


def committed_counters(name, lib_hash, profiles_dir=None):
    """A committed rocprofv3 counter summary (profiles/<name>, written by tools/profile_summary.py) -- but only when
    it was taken on the very build that is being timed: the file names the library's sha256, and a file without one
    or with another one describes some other kernel.  -> (dict or None, note or None)."""
    path = os.path.join(profiles_dir or os.path.join(ROOT, "profiles"), name)
    try:
        with open(path) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None, f"no committed counters ({name})"
    if not lib_hash or t.get("lib_sha256") != lib_hash:
        return None, (f"profiles/{name} was measured on another build of the library (its lib_sha256 "
                      f"{str(t.get('lib_sha256'))[:12]} is not this one's {str(lib_hash)[:12]}): not quoted")
    return t, None


def measured_traffic(lib_hash, profiles_dir=None):
    """HBM bytes per launch of the timed kernel from the committed FETCH_SIZE / WRITE_SIZE passes -> (dict or None, note)."""
    t, note = committed_counters("hbm_traffic.json", lib_hash, profiles_dir)
    if t and not (t.get("fft_size") == N_FFT and t.get("noverlap") == NOVERLAP and t.get("seconds") == SECONDS):
        return None, "profiles/hbm_traffic.json describes another workload"
    return t, note


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def visible_gpus(topology="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs this process could use, counted WITHOUT any HIP call (the launcher process must stay GPU-free: it goes on
    to start children, and torch.cuda.device_count() may initialise the runtime): the KFD topology nodes that have
    SIMDs, cut down by a *_VISIBLE_DEVICES list if one is set.  None when the topology cannot be read."""
    try:
        nodes = os.listdir(topology)
    except OSError:
        return None
    gpus = 0
    for node in nodes:
        try:
            with open(os.path.join(topology, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                gpus += 1
        except (OSError, ValueError):
            continue
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            gpus = min(gpus, len([t for t in v.split(",") if t.strip() != ""]))
    return gpus


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: starts N ranks of this script (one per GPU,
    torch.distributed.run on 127.0.0.1) as child processes (never exec) and returns their exit code.  This process
    makes no HIP call, before or after: it counts GPUs from the KFD topology in /sys and waits for the children; when
    the topology cannot be read it starts the ranks anyway and lets a rank without a device fail loudly."""
    import subprocess
    backend = os.environ.get("SONO_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        have = visible_gpus()
        if have is not None and have < n:
            sys.exit(f"bench.py: --gpus {n} needs {n} visible GPUs, this machine shows {have} "
                     f"(SONO_BENCH_BACKEND=gloo rehearses the multi-rank path on fewer cards)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def measured_valu(lib_hash, profiles_dir=None):
    """Vector-ALU counters per launch of the timed kernel from the committed SQ passes -> (dict or None, note)."""
    return committed_counters("valu_counters.json", lib_hash, profiles_dir)


WORKLOADS = {
    # name: fft, window, alpha, noverlap, sample rate
    "cfg2": dict(fft=1024, window="hann", alpha=None, noverlap=768, sr=384000),
    "cfg3": dict(fft=1024, window="hann", alpha=None, noverlap=768, sr=500000),
    "cfg4": dict(fft=4096, window="blackman", alpha=0.16, noverlap=3891, sr=384000),
    "cfg5": dict(fft=1024, window="hann", alpha=None, noverlap=768, sr=384000),
    # cfg2 on the recording as a WAV file holds it: 16-bit PCM in HBM, converted in the STFT kernel's loads
    "cfg2_pcm16": dict(fft=1024, window="hann", alpha=None, noverlap=768, sr=384000),
    "cfg5_pcm16": dict(fft=1024, window="hann", alpha=None, noverlap=768, sr=384000),
}
FILE_SAMPLES = 2500000          # cfg3: 5 s at 500 kHz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20; cfg5: 2)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps before them (default 3; cfg5: 1)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cfg2")
    ap.add_argument("--seconds", type=int, default=None,
                    help="length of the recording per GPU (default: 3600 for cfg2, 600 for cfg4, 43200 for cfg5)")
    ap.add_argument("--files", type=int, default=10000, help="cfg3: number of files over all ranks")
    ap.add_argument("--no-end-to-end", action="store_true", help="cfg2: skip the host-pointer (PCIe-inclusive) measurement")
    ap.add_argument("--no-latency", action="store_true", help="cfg2: skip the small-call latency figures")
    ap.add_argument("--cpu-seconds", type=float, default=600.0,
                    help="seconds of audio in the CPU baseline's sample, processed repeatedly for >= 10 s (0 disables it)")
    args = ap.parse_args()

    if args.steps is None:
        args.steps = 2 if args.workload == "cfg5" else 20
    if args.warmup is None:
        args.warmup = 1 if args.workload == "cfg5" else 3
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus)
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: start as many ranks as GPUs")

    import torch
    import spectrogram_amd as sa
    from spectrogram_amd import sharding, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # (SONO_BENCH_FORCE_DIST=1 under a one-rank launcher: the process group, barrier and collectives of the multi-rank
    # path over RCCL on a box with one GPU)
    distributed = world > 1 or (os.environ.get("SONO_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    # SONO_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
    backend = os.environ.get("SONO_BENCH_BACKEND", "nccl")
    device_index = local_rank % max(1, torch.cuda.device_count()) if backend != "nccl" else local_rank
    torch.cuda.set_device(device_index if distributed else 0)
    if distributed:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)
    dev = torch.cuda.current_device()
    coll_device = "cuda" if backend == "nccl" else "cpu"

    w = WORKLOADS[args.workload]
    n_fft, noverlap, sr = w["fft"], w["noverlap"], w["sr"]
    hop, bins = n_fft - noverlap, n_fft // 2
    if args.workload == "cfg3":
        my_files = sharding.files_of_rank(args.files, rank, world)
        n_streams, per = len(my_files), FILE_SAMPLES
        seconds = per / sr
    else:
        seconds = args.seconds or {"cfg2": SECONDS, "cfg2_pcm16": SECONDS, "cfg4": 600, "cfg5": 12 * 3600, "cfg5_pcm16": 12 * 3600}[args.workload]
        n_streams, per = 1, int(seconds * sr)
    n = n_streams * per
    frames_per_stream = sa.frames_for(per, n_fft, noverlap)
    frames = n_streams * frames_per_stream
    x = synth.chirp_torch(0, n, sr, stream=rank, device="cuda")          # cfg3: file i = samples [i*per, (i+1)*per)
    engine = sa.SpectrogramEngine(n_fft, w["window"], w["alpha"], device=dev)
    stream = torch.cuda.Stream()          # the stream every launch of the timed region goes to
    peak_bin = torch.empty(frames, dtype=torch.int16, device="cuda")
    peak_mag = torch.empty(frames, dtype=torch.float32, device="cuda")
    gmax = torch.zeros(1, dtype=torch.int32, device="cuda")
    features_only = args.workload in ("cfg5", "cfg5_pcm16")
    pcm16 = args.workload in ("cfg2_pcm16", "cfg5_pcm16", "cfg5")
    if pcm16:
        x = torch.clamp(torch.round(x * 32768.0), -32768, 32767).to(torch.int16)
        torch.cuda.empty_cache()        # the float recording's block goes back to the device: the library allocates beside torch
    bytes_per_frame = (2 if pcm16 else 4) * hop + (0 if features_only else bins) + 6
    detect = args.workload == "cfg5"          # the survey pass with the detector behind it
    if detect:
        # the whole-file detector takes the recording as a host array, like the reference's processFullFile; here as the
        # 16-bit samples a WAV file holds (sono_process_full_file_pcm16): 2 bytes per sample on the host and over the link
        from spectrogram_amd import framing
        x_host = x.cpu().numpy()
        det_rows = []
    if not features_only:
        out = torch.empty(frames * bins, dtype=torch.uint8, device="cuda")
    else:
        # the detector's fast scan (batCallDetector.js:878-949): 50 % overlap, band 10-120 kHz, threshold -60 dB
        scan_v = n_fft // 2
        scan_hop = n_fft - scan_v
        scan_frames = sa.frames_for(per, n_fft, scan_v)
        res = sr / n_fft
        min_bin, max_bin = int(10000.0 // res), min(bins - 1, int(120000.0 // res))
        target = (10.0 ** (-60.0 / 10.0)) * n_fft * 0.25
        active = torch.empty(scan_frames, dtype=torch.uint8, device="cuda")
        run_first = torch.empty(scan_frames // 2 + 1, dtype=torch.int64, device="cuda")
        run_last = torch.empty(scan_frames // 2 + 1, dtype=torch.int64, device="cuda")
        run_count = torch.zeros(1, dtype=torch.int64, device="cuda")

    def step():
        with torch.cuda.stream(stream):
            gmax.zero_()
        if pcm16 and not features_only:
            engine.stft_u8_pcm16_dev(x.data_ptr(), n_streams, per, per, 1, 0, noverlap, GAIN_DB, RANGE_DB, out.data_ptr(),
                                     peak_bin.data_ptr(), peak_mag.data_ptr(), gmax.data_ptr(), stream.cuda_stream)
            return
        if pcm16 and not detect:
            # features at hop N/4 and the fast scan's flags at hop N/2 in ONE read of the samples, then the run list
            engine.stft_survey_pcm16_dev(x.data_ptr(), 1, per, per, 1, 0, min_bin, max_bin, target, peak_bin.data_ptr(),
                                         peak_mag.data_ptr(), gmax.data_ptr(), active.data_ptr(), 0, stream.cuda_stream)
            sa._check(sa.lib().sono_detect_runs_dev(engine._handle(), active.data_ptr(), scan_frames, run_first.data_ptr(),
                                                    run_last.data_ptr(), run_count.data_ptr(), stream.cuda_stream))
            return
        if detect:
            engine.stft_u8_pcm16_dev(x.data_ptr(), 1, per, per, 1, 0, noverlap, GAIN_DB, RANGE_DB, 0, peak_bin.data_ptr(),
                                     peak_mag.data_ptr(), gmax.data_ptr(), stream.cuda_stream)
            # processFullFile (batCallDetector.js:584-840) in one library call: fast scan at 50 % overlap, regions,
            # analysis spectrograms at hop 32, framing and measurement of every call; then the table's rows
            det_rows[:] = [framing.process_full_file_table(engine, x_host, sr, 10.0, 128.0, {"threshold_dB": -60, "padding_ms": 5},
                                                           {"debugMode": True}, file_id=rank, channel=0)]
            return
        if not features_only:
            engine.stft_u8_dev(x.data_ptr(), n_streams, per, per, noverlap, GAIN_DB, RANGE_DB, out.data_ptr(),
                               peak_bin.data_ptr(), peak_mag.data_ptr(), gmax.data_ptr(), stream.cuda_stream)
            return
        engine.stft_features_dev(x.data_ptr(), 1, per, per, noverlap, peak_bin.data_ptr(), peak_mag.data_ptr(),
                                 gmax.data_ptr(), stream.cuda_stream)
        engine.stft_band_active_dev(x.data_ptr(), 1, per, per, scan_v, min_bin, max_bin, target, active.data_ptr(), 0,
                                    stream.cuda_stream)
        sa._check(sa.lib().sono_detect_runs_dev(engine._handle(), active.data_ptr(), scan_frames, run_first.data_ptr(),
                                                run_last.data_ptr(), run_count.data_ptr(), stream.cuda_stream))

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
        stream.synchronize()        # untimed; an engine that sizes its per-XCD ranges by the launch before gets to see it
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

    tmax = torch.tensor([wall], dtype=torch.float64, device=coll_device)
    if distributed:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall_max = float(tmax.item())

    # per-rank summary row, gathered with RCCL (the only collective of the path)
    gmax_f = float(gmax.view(torch.float32).item())
    active_frames = int((peak_mag > 0.4 * gmax_f).sum().item())
    extra = float(run_count.item()) if features_only else 0.0
    call_table = None
    if detect:
        # the path's one collective: the per-GPU call-summary tables, gathered (row counts first, then the rows)
        call_table = sharding.gather_call_tables(det_rows[0], device=coll_device)
        extra = float(det_rows[0].shape[0])
    row = torch.tensor([float(rank), float(frames), gmax_f, float(active_frames), extra], dtype=torch.float64,
                       device=coll_device)
    rows = [row]
    if distributed:
        rows = [torch.empty_like(row) for _ in range(world)]
        dist.all_gather(rows, row)

    if rank == 0:
        total_frames = int(sum(r[1].item() for r in rows))
        ms_per_step = wall_max * 1e3 / args.steps
        kernel_ms = gpu_ms / args.steps
        achieved = frames * bytes_per_frame / (kernel_ms * 1e-3) / 1e9
        flops_per_frame = 5 * n_fft * (n_fft.bit_length() - 1) + n_fft + 16 * bins
        tflops = frames * flops_per_frame / (kernel_ms * 1e-3) / 1e12
        what = {
            "cfg2": f"{seconds:g} s synthetic {sr} Hz mono FM-chirp recording per GPU, FFT {n_fft} Hann, 75% overlap (hop {hop}), "
                    f"uint8 dB frames + per-frame peak bin/magnitude + global max, {frames} frames per GPU, samples resident in HBM",
            "cfg2_pcm16": f"{seconds:g} s synthetic {sr} Hz mono recording per GPU as 16-bit PCM in HBM (converted in the STFT kernel's "
                          f"loads), FFT {n_fft} Hann, 75% overlap (hop {hop}), uint8 dB frames + peaks, {frames} frames per GPU",
            "cfg3": f"{args.files} synthetic 5 s {sr} Hz files ({frames_per_stream} frames each) dealt round-robin to {world} GPU(s), "
                    f"one launch per GPU over its files, FFT {n_fft} Hann, 75% overlap, uint8 dB frames + peaks",
            "cfg4": f"{seconds:g} s synthetic {sr} Hz recording per GPU, FFT {n_fft} Blackman(0.16), 95% overlap (hop {hop}), "
                    f"uint8 dB frames + peaks, {frames} frames per GPU",
            "cfg5_pcm16": f"{seconds / 3600:g} h synthetic {sr} Hz channel per GPU as 16-bit PCM in HBM: per-frame peak features at FFT "
                          f"{n_fft} 75% overlap ({frames} frames, no u8 rows) and the fast scan's flags at 50% overlap from ONE "
                          f"launch over the samples (sono_stft_survey_pcm16_dev) + run extraction",
            "cfg5": f"{seconds / 3600:g} h synthetic {sr} Hz channel per GPU as 16-bit PCM: per-frame peak features at FFT {n_fft} 75% "
                    f"overlap ({frames} frames, no u8 rows) + the whole detector on the channel (processFullFile: fast scan at 50% "
                    f"overlap, regions, analysis spectrograms at hop 32, call framing and measurement; recording handed over as a "
                    f"host int16 array, so its upload is inside the step) + call-summary rows, gathered over the ranks after the "
                    f"timed region",
        }[args.workload]
        result = {
            "metric": "STFT frames/sec (audio-sec/sec) at FFT 1024, 75% overlap, 384 kHz; % HBM roofline",
            "value": total_frames * args.steps / wall_max,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong" if args.workload == "cfg3" else "weak",
            "vs_baseline": None,
            "dtype": "f32",                      # the arithmetic type; cfg2_pcm16 reads int16 and computes in f32
            "data": "synthetic",
            "audio_s_per_s": total_frames * hop / sr * args.steps / wall_max,
            "config": {
                "workload": what, "name": args.workload,
                "fft_size": n_fft, "noverlap": noverlap, "sample_rate": sr, "window": w["window"],
                "frames_per_gpu": frames, "parallelism": f"{world} rank(s), independent units, no data-path collective",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "kernel_ms": kernel_ms, "bytes_per_frame": bytes_per_frame,
                "algorithmic_bytes": frames * bytes_per_frame,
            },
            "valu_nominal": {"achieved": tflops, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / VALU_PEAK_TFLOPS,
                             "flops_per_frame": flops_per_frame,
                             "note": "the conventional flop count of a transform of this size (5 N log2 N + window + magnitude), NOT "
                                     "what the kernel executes (a real-input transform with far fewer operations): a yardstick for "
                                     "comparing with other FFTs, not a utilisation; the measured issue rate is in valu_issue"},
            "summary_rows": [[float(v) for v in r.tolist()] for r in rows],
        }
        if features_only:
            # the step is more than the one kernel the roofline prices: say so instead of a misleading fraction
            result["roofline"]["note"] = ("kernel_ms covers the features pass AND what follows it in the step (fast scan; cfg5: "
                                          "the detector's device work); bytes_per_frame counts the features pass only")
        if detect:
            import resource
            result["calls"] = {"rows_all_ranks": int(call_table.shape[0]), "rows_rank0": int(det_rows[0].shape[0]),
                               "columns": len(sharding.COLUMNS)}
            result["ms_per_hour_of_audio"] = ms_per_step / (seconds / 3600.0)
            result["host_rss_peak_gb"] = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576.0
            result["host_threads_per_rank"] = sa.host_threads()
        lib_hash = library_sha256()
        result["lib_sha256"] = lib_hash
        traffic, counters = None, None
        if args.workload == "cfg2" and seconds == SECONDS:
            traffic, note = measured_traffic(lib_hash)
            if traffic:
                result["roofline"]["traffic"] = traffic["hbm_bytes"]
                result["roofline"]["traffic_source"] = traffic["source"]
            else:
                result["roofline"]["traffic_note"] = note
            counters, note = measured_valu(lib_hash)
            if not counters:
                result["valu_issue"] = None
                result["valu_issue_note"] = note
        if counters:
            # what the vector ALU actually issued (hardware counters of the committed profile), beside the
            # conventional-flop figure above: instructions per launch and how busy that keeps the issue ports
            result["valu_issue"] = {"sq_insts_valu": counters["insts_valu"],
                                    "issue_busy": counters.get("issue_busy_at_2_cycles"),
                                    "instructions_per_frame": counters["insts_valu"] * 64.0 / frames,
                                    "note": "SQ_INSTS_VALU per launch (wave instructions); issue_busy = instructions x 2 cycles / "
                                            "(1024 SIMDs x the dispatch's shader cycles): plain add/mul/fma issue at one per 2 cycles "
                                            "and SIMD, conversions, min/max, selects and DPP moves at 4.8, v_log_f32 at 9 "
                                            "(tools/valubench.hip)",
                                    "source": counters.get("source")}
        if world == 1 and args.workload == "cfg2" and not args.no_end_to_end:
            result["end_to_end"] = end_to_end(sa, torch, engine, x, n_fft, noverlap, frames, bins)
        if world == 1 and args.workload == "cfg2" and not args.no_latency:
            # the application's own call sizes (seconds-long clips, millisecond selections) through the host-pointer
            # entry points, p50 / p95 beside the C oracle: tools/latency_bench.py.  Outside the timed region, never `value`.
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import latency_bench
            result["latency"] = latency_bench.measure(reps=20, quick=True)
        if world == 1 and args.cpu_seconds > 0 and args.workload == "cfg2":
            threads = min(16, os.cpu_count() or 1)
            n_cpu = int(min(args.cpu_seconds, seconds) * sr)
            result["cpu_baseline"] = cpu_baseline(x[:n_cpu].cpu().numpy(), threads)
        print(json.dumps(result), flush=True)

    engine.free()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
