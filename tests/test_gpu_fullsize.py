"""Size-independent properties at the benchmark's full size (1 h at 384 kHz,
FFT 1024, 75 % overlap: 5 399 997 frames), where running the oracle on
everything would take minutes: spot frames against the oracle, exact
amplitude-scaling and time-shift identities, and a checksum that must not
depend on how the work is cut into launches."""
import numpy as np
import pytest

import spectrogram_amd as sa
from oracle import sono_oracle as so
from spectrogram_amd import synth
from tests import parity

pytestmark = pytest.mark.gpu

N, V, H, SR = 1024, 768, 256, 384000
SAMPLES = 3600 * SR


@pytest.fixture(scope="module")
def hour():
    import torch
    x = synth.chirp_torch(0, SAMPLES, SR, device="cuda")
    F = sa.frames_for(SAMPLES, N, V)
    assert F == 5399997
    e = sa.SpectrogramEngine(N, "hann")
    out = torch.empty(F * 512, dtype=torch.uint8, device="cuda")
    pb = torch.empty(F, dtype=torch.int16, device="cuda")
    pm = torch.empty(F, dtype=torch.float32, device="cuda")
    gm = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    # launches go to the engine's own stream (stream=None); synchronise around them
    e.stft_u8_dev(x.data_ptr(), 1, SAMPLES, SAMPLES, V, 20.0, 80.0, out.data_ptr(), pb.data_ptr(), pm.data_ptr(),
                  gm.data_ptr())
    e.synchronize()
    yield dict(x=x, F=F, e=e, out=out.view(F, 512), pb=pb, pm=pm, gm=gm, torch=torch)
    e.free()


def test_spot_frames_against_oracle(hour):
    torch = hour["torch"]
    rng = np.random.default_rng(0)
    # frames inside pulses (every 100 ms = 150 frames) and random ones
    picks = sorted(set(list(rng.integers(0, hour["F"], 48)) + [0, hour["F"] - 1] + [150 * k + 3 for k in (1, 777, 30000)]))
    o = so.OracleEngine(N, "hann")
    nd = nt = 0
    for f in picks:
        xs = hour["x"][f * H: f * H + N].cpu().numpy()
        ref = o.compute_spectrogram_u8(xs, V, 20.0, 80.0)
        got = hour["out"][f].cpu().numpy()
        a, b = parity.u8_diff(got, ref, f"frame {f}")
        nd += a
        nt += b
        assert int(hour["pb"][f].item()) & 0xFFFF == int(o.get_peaks(0.0)[0])
        assert abs(float(hour["pm"][f].item()) - float(o.get_peak_magnitudes(0.0)[0])) <= 1e-6 * o.get_global_max()
    assert nd <= max(2, parity.U8_MAX_FRACTION * nt)


def test_adaptive_ranges_change_nothing(hour, monkeypatch):
    # A launch of this size is cut into one range of tiles per XCD, and the engine sizes the ranges by how fast each XCD
    # got through the launch before (the workgroups report their times); the engine option "fixed_tiles" keeps
    # the ranges equal.  Where the cuts fall must not show anywhere: rows, peaks and the global maximum equal bit for
    # bit, for the whole hour and for a ragged count of frames, launch after launch while the cut adapts.
    torch = hour["torch"]
    for frames in (hour["F"], 1234567):
        n = (frames - 1) * H + N
        res = []
        for fixed in (True, False, False, False):
            hour["e"].set_option("fixed_tiles", fixed)
            out = torch.zeros(frames * 512, dtype=torch.uint8, device="cuda")
            pb = torch.zeros(frames, dtype=torch.int16, device="cuda")
            pm = torch.zeros(frames, dtype=torch.float32, device="cuda")
            gm = torch.zeros(1, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            hour["e"].stft_u8_dev(hour["x"].data_ptr(), 1, n, n, V, 20.0, 80.0, out.data_ptr(), pb.data_ptr(), pm.data_ptr(), gm.data_ptr())
            hour["e"].synchronize()
            res.append((out, pb, pm, gm))
        for other in res[1:]:
            for a, b in zip(res[0], other):
                assert torch.equal(a, b)
        assert torch.equal(res[0][0][: 512 * 1000], hour["out"].reshape(-1)[: 512 * 1000])
        del res


def test_global_max_is_max_of_frame_peaks(hour):
    torch = hour["torch"]
    gmax = hour["gm"].view(torch.float32)[0].item()
    assert gmax == hour["pm"].max().item() and gmax > 0.05


def test_time_shift_identity(hour):
    # The kernel transforms frames in pairs (2q, 2q+1).  The same launch started an even number of
    # hops later keeps every pair together and must reproduce the rows bit for bit; started an odd
    # number of hops later it pairs every frame with its other neighbour, which may move a result by
    # one rounding step but never by more than the u8 tolerance.
    torch = hour["torch"]
    cnt = 200000
    n = (cnt - 1) * H + N
    out = torch.empty(cnt * 512, dtype=torch.uint8, device="cuda")
    for k in (8, 7):
        xs = hour["x"][k * H: k * H + n]
        torch.cuda.synchronize()
        hour["e"].stft_u8_dev(xs.data_ptr(), 1, n, n, V, 20.0, 80.0, out.data_ptr())
        hour["e"].synchronize()
        got, ref = out.view(cnt, 512), hour["out"][k:k + cnt]
        if k % 2 == 0:
            assert torch.equal(got, ref)
        else:
            d = (got.to(torch.int16) - ref.to(torch.int16)).abs()
            assert d.max().item() <= parity.U8_MAX_LSB
            assert (d > 0).sum().item() <= parity.U8_MAX_FRACTION * d.numel()


def test_launch_geometry_independent_checksum(hour):
    # a sum over all u8 codes, computed from chunked launches with halo, equals the one-launch sum
    # (chunks start on even frame numbers, which keeps the kernel's frame pairs together)
    torch = hour["torch"]
    F = hour["F"]
    total = hour["out"].to(torch.int64).sum().item() if F < 1000 else sum(
        hour["out"][i:i + 500000].sum(dtype=torch.int64).item() for i in range(0, F, 500000))
    acc = 0
    step = 1333334
    buf = torch.empty(step * 512, dtype=torch.uint8, device="cuda")
    for f0 in range(0, F, step):
        f1 = min(f0 + step, F)
        n = (f1 - f0 - 1) * H + N
        xs = hour["x"][f0 * H: f0 * H + n]
        torch.cuda.synchronize()
        hour["e"].stft_u8_dev(xs.data_ptr(), 1, n, n, V, 20.0, 80.0, buf.data_ptr())
        hour["e"].synchronize()
        acc += buf[: (f1 - f0) * 512].sum(dtype=torch.int64).item()
    assert acc == total


def test_amplitude_scaling_identity(hour):
    # scaling the input by 2^-3 scales every linear magnitude by exactly 2^-3
    torch = hour["torch"]
    cnt = 100000
    n = (cnt - 1) * H + N
    xs = hour["x"][:n].clone()
    a = torch.empty(cnt * 512, dtype=torch.float32, device="cuda")
    b = torch.empty(cnt * 512, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    hour["e"].stft_mag_f32_dev(xs.data_ptr(), 1, n, n, V, a.data_ptr())
    hour["e"].synchronize()
    xs.mul_(0.125)
    torch.cuda.synchronize()
    hour["e"].stft_mag_f32_dev(xs.data_ptr(), 1, n, n, V, b.data_ptr())
    hour["e"].synchronize()
    assert torch.equal(a * 0.125, b)


def test_pulse_frames_peak_on_the_sweep(hour):
    # frames centred inside a pulse must peak between 30 and 80 kHz (bins 80..214 at 375 Hz/bin)
    torch = hour["torch"]
    pb = hour["pb"].to(torch.int32) & 0xFFFF
    for pulse in (0, 5, 12345, 35999):
        f = pulse * 150 + 2            # frame starting 2 hops into the pulse
        assert 75 <= pb[f].item() <= 220, (pulse, pb[f].item())


def test_peak_bins_against_first_maximum_of_the_magnitude_rows(hour):
    """The reference takes the per-frame peak on the ROUNDED f32 magnitude, first maximum wins (lib.rs:370-378); the
    kernel takes it on |2X|^2 -- largest power, lowest bin among equal powers -- and the two differ exactly when two
    bins whose powers differ by an ulp or two round to the same magnitude and the earlier one has the smaller power.
    Counted here on the benchmark recording: over every frame, the kernel's peak bin against the first maximum of
    its own magnitude rows (sqrt of the same powers).  Such near-ties are rare enough to be counted in single
    frames, and where they happen the two bins hold the same magnitude bit for bit, so peak magnitudes, the global
    maximum and every threshold test are unaffected."""
    torch = hour["torch"]
    e, x, F = hour["e"], hour["x"], hour["F"]
    block = 1 << 19
    mags = torch.empty(block * 512, dtype=torch.float32, device="cuda")
    differ = ties = 0
    for f0 in range(0, F, block):
        nf = min(block, F - f0)
        ns = (nf - 1) * H + N
        e.stft_mag_f32_dev(x.data_ptr() + 4 * f0 * H, 1, ns, ns, V, mags.data_ptr())
        e.synchronize()
        m = mags[: nf * 512].view(nf, 512)
        top = m.max(dim=1, keepdim=True).values
        at_top = m == top
        first = torch.where(at_top, torch.arange(512, device="cuda").expand(nf, 512), 512).min(dim=1).values
        got = hour["pb"][f0:f0 + nf].to(torch.int64) & 0xFFFF
        silent = top[:, 0] == 0
        bad = (first != got) & ~silent
        differ += int(bad.sum().item())
        ties += int(((at_top.sum(dim=1) > 1) & ~silent).sum().item())
        assert torch.equal(hour["pm"][f0:f0 + nf], top[:, 0])            # the peak magnitude is the row maximum, always
        if bad.any():                                                   # where they differ, both bins hold the row maximum
            idx = torch.nonzero(bad)[:, 0]
            assert torch.equal(m[idx, got[idx]], top[idx, 0])
    print(f"frames {F}: magnitude ties at the row maximum {ties}, peak bin differs from the first maximum in {differ}")
    # Measured on this recording: 7 frames hold a tie at the row maximum, in 3 of them the kernel reports the later bin.
    # The bound is that count plus a margin for the last-bit differences between builds, not a rate.
    assert ties <= 16 and differ <= 8
