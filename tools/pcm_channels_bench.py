import sys, torch, numpy as np
sys.path.insert(0, '.')
import spectrogram_amd as sa
N, V = 1024, 512
per = 384000 * 600
for ch in (1, 2, 3):
    pcm = (torch.randn(per * ch, device="cuda") * 3000).to(torch.int16)
    e = sa.SpectrogramEngine(N, "hann")
    F = sa.frames_for(per, N, V)
    act = torch.zeros(F, dtype=torch.uint8, device="cuda")
    st = torch.cuda.Stream()
    for it in range(3):
        e.stft_band_active_pcm16_dev(pcm.data_ptr(), 1, per, per, ch, ch - 1, V, 10, 400, 1.0, act.data_ptr(), 0, st.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for it in range(10):
        e.stft_band_active_pcm16_dev(pcm.data_ptr(), 1, per, per, ch, ch - 1, V, 10, 400, 1.0, act.data_ptr(), 0, st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    print(f"channels {ch}: fast scan of 10 min: {e0.elapsed_time(e1) / 10:.3f} ms")
    e.free()
    del pcm
