// sonoradar_gpu.js -- the module shell SonoRadar imports in place of
// modules/spectrogram_wasm.js when it runs under Node on a GPU box: same export
// names as the wasm-bindgen glue (modules/spectrogram_wasm.d.ts:4-335), backed by
// the N-API addon sonostft.node -> libsonostft.so -> HIP kernels.
//
//   const wasm = require('./napi/sonoradar_gpu.js');
//   await wasm.default();                       // init(): resolves at once, nothing to fetch or compile
//   const e = new wasm.SpectrogramEngine(1024, 'hann', null);
//
// There is no CPU fallback here: without the addon or without a GPU the
// constructor throws, which is what batCallDetector's own wasm -> JS fallback
// (batCallDetector.js:858-870) keys on.
'use strict';
const path = require('path');

// SONO_ADDON_PATH: another build of the addon (the sanitizer build of tools/asan_tests.sh)
const addon = require(process.env.SONO_ADDON_PATH || path.join(__dirname, 'sonostft.node'));

class SpectrogramEngine extends addon.SpectrogramEngine {}
class WaveformEngine extends addon.WaveformEngine {}
if (typeof Symbol.dispose === 'symbol') {
  SpectrogramEngine.prototype[Symbol.dispose] = SpectrogramEngine.prototype.free;
  WaveformEngine.prototype[Symbol.dispose] = WaveformEngine.prototype.free;
}

// init(module_or_path?) / initSync(module): the wasm glue instantiates the module here
// (spectrogram_wasm.d.ts:324-334); the addon is ready as soon as it is loaded.
function initSync() { return module.exports; }
function init() { return Promise.resolve(module.exports); }

// accelerateBatCallDetector(BatCallDetector, BatCall?): replaces the detector's whole-file pass
// (BatCallDetector.prototype.processFullFile, batCallDetector.js:584-840) by ONE native call on the detector's own
// engine (`new BatCallDetector(config, engine)`, js:346-351): fast scan, merge & pad, every region's analysis
// spectrogram, framing and the measurement stage run batched on the GPU and on host threads.  Same calls, same order;
// each call is a BatCall (when the class is given) carrying the numeric fields, callType, quality and -- unless
// options.keepSpectrograms is false -- the call's spectrogram rows, timeFrames and freqBins (js:785-787; they cost
// frames x bins floats of host memory per call).  Rejects where the reference's promise
// rejects; detectors without a GPU engine keep the JavaScript path.  getTimeExpansionMode is fileState.js's getter
// (fileState.js:164-166), which the reference's measurement stage reads as a module global (batCallDetector.js:3891).
function accelerateBatCallDetector(BatCallDetector, BatCall, getTimeExpansionMode) {
  const original = BatCallDetector.prototype.processFullFile;
  BatCallDetector.prototype.processFullFile = async function (fullAudioData, sampleRate, flowKHz, fhighKHz, options = {}) {
    const engine = this.wasmEngine;
    if (!(engine instanceof addon.SpectrogramEngine)) return original.call(this, fullAudioData, sampleRate, flowKHz, fhighKHz, options);
    const progress = options.progressCallback || (() => {});
    const config = Object.assign({}, this.config, { debugMode: !!this.debugMode, timeExpansion: !!(getTimeExpansionMode && getTimeExpansionMode()),
      keepSpectrograms: options.keepSpectrograms !== false });
    const rows = engine.process_full_file(fullAudioData, sampleRate, flowKHz, fhighKHz, options, config);
    progress(1);
    return rows.map((row) => Object.assign(BatCall ? new BatCall() : {}, row));
  };
  return BatCallDetector;
}

module.exports = {
  default: init,
  initSync,
  SpectrogramEngine,
  WaveformEngine,
  compute_power_spectrum: addon.compute_power_spectrum,
  find_peak_frequency_from_spectrum: addon.find_peak_frequency_from_spectrum,
  compute_wave_peaks: addon.compute_wave_peaks,
  find_global_max: addon.find_global_max,
  accelerateBatCallDetector,                 // not in the wasm surface: the detector's whole-file pass as one native call
  device_count: addon.device_count,          // not in the wasm surface: lets a batch driver size its worker pool
};
