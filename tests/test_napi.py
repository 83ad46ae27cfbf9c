"""The Node N-API addon (napi/): its export surface on any machine, and on a GPU
its results against the Python binding of the same library and the oracle."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ADDON = os.path.join(ROOT, "napi", "sonostft.node")

# The export surface of the reference's wasm-bindgen module (modules/spectrogram_wasm.d.ts:4-271, 324-334).
MODULE_EXPORTS = ["default", "initSync", "SpectrogramEngine", "WaveformEngine", "compute_power_spectrum",
                  "find_peak_frequency_from_spectrum", "compute_wave_peaks", "find_global_max"]
EXTRA_EXPORTS = ["accelerateBatCallDetector", "device_count"]         # beyond the wasm surface
ENGINE_METHODS = ["free", "get_fft_size", "get_freq_bins", "get_num_filters", "get_window_values", "load_filter_bank",
                  "clear_filter_bank", "compute_spectrogram", "compute_spectrogram_u8", "get_peaks",
                  "get_peak_magnitudes", "get_global_max", "set_color_map", "set_spectrum_config",
                  "compute_spectrogram_image", "release_memory"]
WAVEFORM_METHODS = ["free", "resize", "load_channel", "get_peaks_in_range", "get_channel_length", "get_num_channels",
                    "clear", "release_memory"]


def need_node():
    if shutil.which("node") is None:
        pytest.skip("node is not installed")
    if not os.path.exists(ADDON):
        from napi import build_addon
        if build_addon.build() is None:
            pytest.skip("Node headers are not installed: the addon cannot be built here")


def node(script):
    return subprocess.check_output(["node", "-e", script], cwd=ROOT, stderr=subprocess.STDOUT, timeout=120).decode()


def test_addon_presents_the_wasm_module_surface():
    need_node()
    out = json.loads(node("""
        const w = require('./napi/sonoradar_gpu.js');
        const own = (c) => Object.getOwnPropertyNames(Object.getPrototypeOf(c.prototype));
        const x = new Float32Array([0.1, -0.5, 0.3, 0.2]);
        const we = new w.WaveformEngine(); we.resize(2); we.load_channel(1, x);
        const r = {exports: Object.keys(w), engine: own(w.SpectrogramEngine), wave: own(w.WaveformEngine),
                   gmax: w.find_global_max(x), peaks: Array.from(w.compute_wave_peaks(x, 2)),
                   chans: we.get_num_channels(), len: [we.get_channel_length(0), we.get_channel_length(1)],
                   range: Array.from(we.get_peaks_in_range(1, 0, 4, 2)),
                   peakfreq: w.find_peak_frequency_from_spectrum(new Float32Array([0, 1, 5, 2, 0]), 8000, 8, 0, 4000)};
        we.free();
        try { we.clear(); } catch (e) { r.after_free = e.message; }
        w.default().then((m) => { r.init_ok = m === w && w.initSync() === w; console.log(JSON.stringify(r)); });
    """))
    assert set(MODULE_EXPORTS + EXTRA_EXPORTS) == set(out["exports"]) and "process_full_file" in out["engine"]
    assert set(ENGINE_METHODS) <= set(out["engine"]) and set(WAVEFORM_METHODS) <= set(out["wave"])
    assert out["init_ok"] and out["gmax"] == 0.5 and out["peaks"] == [0.5, np.float32(0.3)]
    assert out["chans"] == 2 and out["len"] == [0, 4] and out["range"] == [0.5, np.float32(0.3)]
    assert out["after_free"] == "null pointer passed to rust"
    assert abs(out["peakfreq"] - (2 + 1 / 14) * 1000) < 1e-3     # parabola through (1, 5, 2) dB around bin 2 of 1000 Hz bins (lib.rs:1163-1172)


def test_addon_without_gpu_throws_instead_of_falling_back():
    need_node()
    import spectrogram_amd as sa
    if sa.device_count() > 0:
        pytest.skip("a GPU is present")
    out = node("""
        const w = require('./napi/sonoradar_gpu.js');
        try { new w.SpectrogramEngine(1024, 'hann', null); console.log('constructed'); } catch (e) { console.log('threw: ' + e.message); }
    """)
    assert out.startswith("threw: ") and "no HIP device" in out


@pytest.mark.gpu
def test_addon_results_equal_python_binding_and_oracle(tmp_path):
    need_node()
    import spectrogram_amd as sa
    from oracle import sono_oracle as so
    from tests import parity
    from tests.test_gpu_parity import signal
    N, V, sr = 1024, 768, 384000
    x = signal(N + 40 * 256 + 9, 123)
    x.tofile(tmp_path / "x.f32")
    job = dict(input=str(tmp_path / "x.f32"), fft=N, window="hann", noverlap=V, gain=20.0, range=80.0, ratio=0.4, sr=sr,
               out=str(tmp_path))
    (tmp_path / "job.json").write_text(json.dumps(job))
    subprocess.check_call(["node", os.path.join(ROOT, "napi", "run_case.js"), str(tmp_path / "job.json")], timeout=120)
    meta = json.loads((tmp_path / "meta.json").read_text())
    rd = lambda name, dt: np.fromfile(tmp_path / name, dtype=dt)
    e = sa.SpectrogramEngine(N, "hann")
    o = so.OracleEngine(N, "hann")
    assert meta["fft"] == N and meta["bins"] == 512 and meta["filters"] == 0 and meta["peaks_before"] == 0
    assert np.array_equal(rd("window.f32", np.float32), e.get_window_values())
    assert np.array_equal(rd("mag.f32", np.float32), e.compute_spectrogram(x, V))           # same library: bit-exact
    assert np.array_equal(rd("u8.u8", np.uint8), e.compute_spectrogram_u8(x, V, 20.0, 80.0))
    assert np.array_equal(rd("peaks.u16", np.uint16), e.get_peaks(0.4))
    assert np.array_equal(rd("peakmags.f32", np.float32), e.get_peak_magnitudes(0.4))
    assert meta["gmax"] == e.get_global_max()
    parity.check_mag(rd("mag.f32", np.float32), o.compute_spectrogram(x, V), 512, "addon mag")
    nd, nt = parity.u8_diff(rd("u8.u8", np.uint8), o.compute_spectrogram_u8(x, V, 20.0, 80.0), "addon u8")
    assert nd <= max(2, parity.U8_MAX_FRACTION * nt)
    assert np.array_equal(rd("peaks.u16", np.uint16), o.get_peaks(0.4))
    spec = rd("power.f32", np.float32)
    parity.check_power(spec, so.compute_power_spectrum(x, sr, N, "hann", 75.0), "addon power spectrum")
    assert meta["peakfreq"] == sa.find_peak_frequency_from_spectrum(spec, sr, N, 10000, 120000)
    assert "noverlap must be < fft_size" in meta["noverlap_error"]        # the reference panics here (lib.rs:118-120)
    assert meta["after_free"] == "null pointer passed to rust"
    e.free()


@pytest.mark.gpu
def test_addon_whole_file_pass_equals_python_binding(tmp_path):
    """engine.process_full_file and the accelerateBatCallDetector shell against framing.process_full_file_native."""
    need_node()
    import spectrogram_amd as sa
    from spectrogram_amd import framing as fr
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_framing.npz"))
    meta = json.loads(bytes(g["__meta__"]).decode())
    (c,) = [c for c in meta if c["kind"] == "fullfile_measured"]
    x = np.ascontiguousarray(g[c["name"] + "/x"], np.float32)
    x.tofile(tmp_path / "x.f32")
    script = """
        const fs = require('fs');
        const w = require('./napi/sonoradar_gpu.js');
        const buf = fs.readFileSync(%r);
        const x = new Float32Array(buf.buffer, buf.byteOffset, buf.length / 4);
        const e = new w.SpectrogramEngine(%d, %r, null);
        const cfg = {hopPercent: %r, callThreshold_dB: %r, minCallDuration_ms: %r};
        class BatCall { constructor() { this.callType = 'FM'; } }
        class BatCallDetector {
          constructor(config, engine) { this.config = config; this.wasmEngine = engine; this.debugMode = false; }
          async processFullFile() { return 'javascript path'; }
        }
        w.accelerateBatCallDetector(BatCallDetector, BatCall);
        (async () => {
          const r = {};
          const d = new BatCallDetector(cfg, e);
          try { await d.processFullFile(x, %d, %r, %r, {threshold_dB: %r, padding_ms: %r}); } catch (err) { r.rejected = err.message; }
          d.debugMode = true;
          let seen = 0;
          const calls = await d.processFullFile(x, %d, %r, %r, {threshold_dB: %r, padding_ms: %r, progressCallback: (p) => { seen = p; }});
          r.calls = calls.map((k) => Object.assign({}, k, {spectrogram: k.spectrogram.map((row) => Array.from(row)),
                                                           freqBins: Array.from(k.freqBins)}));
          r.shared = calls.every((k) => k.freqBins === calls[0].freqBins && k.spectrogram[0] instanceof Float32Array);
          r.plain = (await d.processFullFile(x, %d, %r, %r, {threshold_dB: %r, padding_ms: %r, keepSpectrograms: false}))
                      .every((k) => k.spectrogram === undefined);
          r.progress = seen; r.isBatCall = calls.every((k) => k instanceof BatCall);
          r.fallback = await new BatCallDetector(cfg, null).processFullFile(x, 1, 1, 1);
          e.free();
          console.log(JSON.stringify(r));
        })();
    """ % (str(tmp_path / "x.f32"), c["fft"], c["window"], c["hop_percent"], c["call_threshold_db"], c["min_call_duration_ms"],
           c["sr"], c["flow_khz"], c["fhigh_khz"], c["scan_threshold_db"], c["padding_ms"],
           c["sr"], c["flow_khz"], c["fhigh_khz"], c["scan_threshold_db"], c["padding_ms"],
           c["sr"], c["flow_khz"], c["fhigh_khz"], c["scan_threshold_db"], c["padding_ms"])
    out = json.loads(node(script))
    assert "outside its debug mode" in out["rejected"] and out["progress"] == 1 and out["isBatCall"]
    assert out["fallback"] == "javascript path"
    e = sa.SpectrogramEngine(c["fft"], c["window"])
    want = fr.process_full_file_native(e, x, c["sr"], c["flow_khz"], c["fhigh_khz"],
                                       {"threshold_dB": c["scan_threshold_db"], "padding_ms": c["padding_ms"]},
                                       {"hopPercent": c["hop_percent"], "callThreshold_dB": c["call_threshold_db"],
                                        "minCallDuration_ms": c["min_call_duration_ms"], "debugMode": True},
                                       keep_spectrograms=True)
    e.free()
    assert len(out["calls"]) == len(want) > 0 and out["shared"] and out["plain"]
    for got, ref in zip(out["calls"], want):
        for k, v in ref.items():
            if k in ("spectrogram", "timeFrames", "freqBins"):           # BatCall's matrix fields, js:785-787
                assert np.array_equal(np.asarray(got[k], v.dtype), v), k
                continue
            assert got[k] == v, (k, got[k], v)                           # same library, doubles through JSON: bit-exact
        assert got["kneeTime_ms"] == ref["kneeFreq_ms"]


@pytest.mark.gpu
def test_addon_takes_a_16_bit_recording_and_engine_options(tmp_path):
    """engine.process_full_file on the Int16Array of a decoded 16-bit WAV channel gives the calls of the Float32Array of
    the same samples / 32768; engine.set_option reaches the engine's switches and rejects unknown names."""
    need_node()
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_framing.npz"))
    meta = json.loads(bytes(g["__meta__"]).decode())
    (c,) = [c for c in meta if c["kind"] == "fullfile_measured"]
    q = np.clip(np.round(np.asarray(g[c["name"] + "/x"], np.float64) * 32767.0), -32768, 32767).astype(np.int16)
    q.tofile(tmp_path / "x.i16")
    script = """
        const fs = require('fs');
        const w = require('./napi/sonoradar_gpu.js');
        const buf = fs.readFileSync(%r);
        const q = new Int16Array(buf.buffer, buf.byteOffset, buf.length / 2);
        const x = Float32Array.from(q, (v) => v / 32768);
        const e = new w.SpectrogramEngine(%d, %r, null);
        const cfg = {hopPercent: %r, callThreshold_dB: %r, minCallDuration_ms: %r, debugMode: 1};
        const opt = {threshold_dB: %r, padding_ms: %r};
        const a = e.process_full_file(q, %d, %r, %r, opt, cfg), b = e.process_full_file(x, %d, %r, %r, opt, cfg);
        const r = {n: a.length, same: JSON.stringify(a) === JSON.stringify(b)};
        e.set_option('force_any_hop', true);
        r.again = JSON.stringify(e.process_full_file(q, %d, %r, %r, opt, cfg)) === JSON.stringify(a);
        try { e.set_option('no_such_switch', true); } catch (err) { r.unknown = err.message; }
        try { e.process_full_file(new Uint8Array(8), 1, 1, 1); } catch (err) { r.wrongType = err.message; }
        e.free();
        console.log(JSON.stringify(r));
    """ % (str(tmp_path / "x.i16"), c["fft"], c["window"], c["hop_percent"], c["call_threshold_db"], c["min_call_duration_ms"],
           c["scan_threshold_db"], c["padding_ms"], c["sr"], c["flow_khz"], c["fhigh_khz"], c["sr"], c["flow_khz"], c["fhigh_khz"],
           c["sr"], c["flow_khz"], c["fhigh_khz"])
    out = json.loads(node(script))
    assert out["n"] > 0 and out["same"] and out["again"]
    assert "unknown option" in out["unknown"] and "Int16Array" in out["wrongType"]
