// run_case.js -- drives the addon through the wasm-module surface for the tests:
//   node run_case.js <job.json>      (job: {input: f32 file, fft, window, noverlap, gain, range, ratio, sr, out: dir})
// Writes mag.f32, u8.u8, peaks.u16, peakmags.f32, power.f32 and meta.json into job.out.
'use strict';
const fs = require('fs');
const path = require('path');
const wasm = require('./sonoradar_gpu.js');

const job = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
const buf = fs.readFileSync(job.input);
const x = new Float32Array(buf.buffer, buf.byteOffset, buf.length / 4);
const put = (name, typed) => fs.writeFileSync(path.join(job.out, name), Buffer.from(typed.buffer, typed.byteOffset, typed.byteLength));

wasm.default().then((m) => {
  const e = new m.SpectrogramEngine(job.fft, job.window, null);
  const meta = { fft: e.get_fft_size(), bins: e.get_freq_bins(), filters: e.get_num_filters() };
  put('window.f32', e.get_window_values());
  meta.peaks_before = e.get_peaks(0.5).length;
  put('mag.f32', e.compute_spectrogram(x, job.noverlap));
  put('u8.u8', e.compute_spectrogram_u8(x, job.noverlap, job.gain, job.range));
  put('peaks.u16', e.get_peaks(job.ratio));
  put('peakmags.f32', e.get_peak_magnitudes(job.ratio));
  meta.gmax = e.get_global_max();
  const spec = m.compute_power_spectrum(x, job.sr, job.fft, job.window, 75);
  put('power.f32', spec);
  meta.peakfreq = m.find_peak_frequency_from_spectrum(spec, job.sr, job.fft, 10000, 120000);
  let threw = '';
  try { e.compute_spectrogram(x, job.fft); } catch (err) { threw = err.message; }
  meta.noverlap_error = threw;
  e.free();
  threw = '';
  try { e.get_fft_size(); } catch (err) { threw = err.message; }
  meta.after_free = threw;
  fs.writeFileSync(path.join(job.out, 'meta.json'), JSON.stringify(meta));
});
