// Test infrastructure only: captures golden vectors of the detector's call-framing
// stage from the reference's own JavaScript.  The methods live in
// /root/reference/modules/batCallDetector.js, a module that cannot be loaded as a
// whole under the Node 12 of the build container; this script reads the file as
// text at run time, cuts the needed method bodies out by name, wraps each in a
// Function and runs them against the reference wasm engine (wasm_ref.js).
// detectCalls itself is run with stand-ins for the three things it calls that are
// outside this stage (the BatCall record, measureFrequencyParameters, the call
// classifier), so its own control flow -- duration filter, padding, oscillogram
// refinement, slicing, noise-floor percentile -- is the reference's, executed.
// Nothing of the reference is stored in this repository: only inputs and outputs.
// usage: node --experimental-wasm-mv --experimental-wasm-anyref framing_driver.js <wasm> <detector.js> <workdir>
'use strict';
const fs = require('fs');
const path = require('path');
const { load } = require('./wasm_ref.js');

const wasmPath = process.argv[2], jsPath = process.argv[3], work = process.argv[4];
const ref = load(wasmPath);
const src = fs.readFileSync(jsPath, 'utf8');
const AsyncFunction = Object.getPrototypeOf(async function () {}).constructor;

// "  [async ]name(args) {" at class-member indentation -> Function(argNames..., body)
function method(name, argNames) {
  const re = new RegExp('\\n  (async )?' + name + '\\(([^)]*)\\) \\{');
  const m = re.exec(src);
  if (!m) throw new Error('method not found: ' + name);
  let i = m.index + m[0].length, depth = 1;
  const start = i;
  for (; i < src.length && depth > 0; i++) {
    if (src[i] === '{') depth++;
    else if (src[i] === '}') depth--;
  }
  const names = argNames || m[2].split(',').map((s) => s.trim());     // defaults stay ("order = 4")
  const Ctor = m[1] ? AsyncFunction : Function;
  return new Ctor(...names, src.slice(start, i - 1));
}
const names = ['fastScanSegmentsWasm', 'mergeAndPadSegments', 'generateSpectrogramWasm', 'detectCallSegments', 'savitzkyGolay', 'refineEndUsingOscillogram',
  'calculateZonalNoiseFloors', '_scanSpectrumForFrequency', 'calculateAutoHighpassFilterFreq', 'applyHighpassFilter',
  '_applyButterworthStage', 'getQualityRating', 'calculateDuration', 'calculateRMSbasedSNR',
  'findOptimalLowFrequencyThreshold', 'findOptimalHighFrequencyThreshold', 'calculateBandwidth'];
const fns = {};
for (const n of names) fns[n] = method(n);
// CallTypeClassifier.classify (batCallDetector.js:48-56), a static method
const classifyCall = (() => {
  const m = /\n  static classify\(call\) \{/.exec(src);
  let i = m.index + m[0].length, depth = 1; const start = i;
  for (; i < src.length && depth > 0; i++) { if (src[i] === '{') depth++; else if (src[i] === '}') depth--; }
  return new Function('call', src.slice(start, i - 1));
})();
fns.detectCalls = method('detectCalls', ['audioData', 'sampleRate', 'flowKHz', 'fhighKHz', 'options']);
// STEP 0 and STEP 0.5 of measureFrequencyParameters (peak bin, parabolic peak frequency, peak time; rejection of
// vertical clicks by the peak frame's instantaneous bandwidth): the method's text up to the comment that opens
// STEP 1; that prefix is a complete sequence of statements.
const peakStep = (() => {
  const m = /\n  measureFrequencyParameters\(([^)]*)\) \{/.exec(src);
  const start = m.index + m[0].length;
  const cut = src.indexOf('[STEP 1]', start);
  if (cut < 0) throw new Error('STEP 1 marker not found');
  const end = src.lastIndexOf('// ====', cut);            // the banner line above the marker
  return new Function(...m[1].split(',').map((t) => t.trim()), src.slice(start, end));
})();

// The whole measureFrequencyParameters, with the one free identifier of its text (the time-expansion switch of
// fileState.js, off by default) supplied as a leading parameter.
const realMeasure = (() => {
  const m = /\n  measureFrequencyParameters\(([^)]*)\) \{/.exec(src);
  let i = m.index + m[0].length, depth = 1; const start = i;
  for (; i < src.length && depth > 0; i++) { if (src[i] === '{') depth++; else if (src[i] === '}') depth--; }
  const f = new Function('getTimeExpansionMode', ...m[1].split(',').map((t) => t.trim()), src.slice(start, i - 1));
  return function (...args) { return f.call(this, () => false, ...args); };
})();
const CALL_FIELDS = ['startTime_s', 'endTime_s', 'duration_ms', 'peakFreq_kHz', 'peakFreq_ms', 'peakPower_dB', 'highFreq_kHz', 'highFreq_ms',
  'highFreqFrameIdx', 'startFreq_kHz', 'startFreq_ms', 'startFreqFrameIdx', 'endFreq_kHz', 'endFreq_ms', 'endFrameIdx_forLowFreq',
  'lowFreq_kHz', 'lowFreq_ms', 'characteristicFreq_kHz', 'characteristicFreq_ms', 'kneeFreq_kHz', 'kneeFreq_ms', 'kneeFrameIdx',
  'heelFreq_kHz', 'heelFreq_ms', 'heelFrameIdx', 'bandwidth_kHz', 'highFreqThreshold_dB_used', 'lowFreqThreshold_dB_used', 'snr_dB',
  'startFreqTime_s', 'endFreqTime_s'];
const callRow = (k) => CALL_FIELDS.map((f) => (k[f] === null || k[f] === undefined ? NaN : k[f]));
function newBatCall() {
  const o = {};
  for (const f of ['startTime_s', 'endTime_s', 'duration_ms', 'peakFreq_kHz', 'peakFreq_ms', 'highFreq_kHz', 'highFreq_ms', 'highFreqFrameIdx',
    'startFreq_kHz', 'startFreq_ms', 'startFreqFrameIdx', 'startFreqTime_s', 'endFreq_kHz', 'endFreq_ms', 'endFreqTime_s', 'lowFreq_kHz',
    'lowFreq_ms', 'lowFreqFrameIdx', 'endFrameIdx_forLowFreq', 'characteristicFreq_kHz', 'characteristicFreq_ms', 'kneeFreq_kHz',
    'kneeFreq_ms', 'kneeTime_ms', 'heelFreq_kHz', 'heelFreq_ms', 'heelFrameIdx', 'bandwidth_kHz', 'peakPower_dB', 'snr_dB',
    'highFreqThreshold_dB_used', 'lowFreqThreshold_dB_used']) o[f] = null;          // the fields BatCall's constructor sets to null
  return o;
}

const cases = JSON.parse(fs.readFileSync(path.join(work, 'cases.json'), 'utf8'));
const inBuf = fs.readFileSync(path.join(work, 'inputs.bin'));
const inputs = new Float32Array(inBuf.buffer, inBuf.byteOffset, inBuf.length / 4);
const outFd = fs.openSync(path.join(work, 'outputs.bin'), 'w');
let outPos = 0;
const manifest = [];
function emit(caseName, key, typed) {
  const b = Buffer.from(typed.buffer, typed.byteOffset, typed.byteLength);
  fs.writeSync(outFd, b);
  const dtype = typed instanceof Float32Array ? 'f32' : typed instanceof Float64Array ? 'f64' : 'u8';
  manifest.push({ case: caseName, key, dtype, offset: outPos, count: typed.length });
  outPos += b.length;
}
const f64 = (a) => Float64Array.from(a);
const zoneArray = (map) => {                       // {"0": dB, "10": dB, ...} -> dB by zone index
  const keys = Object.keys(map).map(Number).sort((a, b) => a - b);
  return f64(keys.map((k) => map[String(k)]));
};
console.log = () => {}; console.warn = () => {}; console.groupCollapsed = () => {}; console.table = () => {}; console.groupEnd = () => {};

async function run() {
  for (const c of cases) {
    const x = inputs.subarray(c.x_off, c.x_off + c.x_len);
    const e = new ref.Engine(c.fft, c.window, null);
    const captured = [];
    const self = {
      debugMode: false,
      wasmEngine: { get_fft_size: () => c.fft, get_freq_bins: () => c.fft / 2, compute_spectrogram: (a, v) => e.mag(a, v) },
      config: { hopPercent: c.hop_percent, callThreshold_dB: c.call_threshold_db, fftSize: c.fft,
        minCallDuration_ms: c.min_call_duration_ms },
    };
    for (const n of names) self[n] = fns[n];
    self.generateSpectrogram = fns.generateSpectrogramWasm;
    self.fastScanSegments = fns.fastScanSegmentsWasm;
    // stand-in for the measurement stage: records what it is handed and leaves a peak power behind
    if (c.debug_mode) self.debugMode = true;
    let measureThrew = 0;
    self.measureFrequencyParameters = c.real_measure ? function (...args) {
      try { return realMeasure.apply(this, args); } catch (err) { measureThrew++; throw err; }
    } : function (call, flowKHz, fhighKHz, freqBins, freqResolution, zonalNoiseMap) {
      let peak = -Infinity;
      for (const row of call.spectrogram) for (let b = 0; b < row.length; b++) if (row[b] > peak) peak = row[b];
      call.peakPower_dB = peak;
      call.lowFreq_kHz = 0; call.highFreq_kHz = 0;
      peakStep.call(self, call, flowKHz, fhighKHz, freqBins, freqResolution, zonalNoiseMap);      // the reference's own STEP 0 / 0.5
      if (call.peakPower_dB !== peak) throw new Error('stand-in and STEP 0 disagree on the peak power');
      captured.push({ frames: call.spectrogram.length, zonal: zoneArray(zonalNoiseMap), peakFreq_kHz: call.peakFreq_kHz,
        peakFreq_ms: call.peakFreq_ms, peakFrameIdx: call.peakFrameIdx, discarded: call.isDiscarded ? 1 : 0 });
    };

    function BatCall() { Object.assign(this, newBatCall()); }
    BatCall.prototype.calculateDuration = fns.calculateDuration;
    BatCall.prototype.calculateBandwidth = function () {          // BatCall.calculateBandwidth (batCallDetector.js:166-170), run as text below
      fns.calculateBandwidth.call(this);
    };
    const CallTypeClassifier = { classify: c.real_measure ? classifyCall : () => 'NA' };
    // a method whose text refers to BatCall and CallTypeClassifier as free identifiers gets them as leading parameters
    const scoped = (name, argNames) => {
      const m = new RegExp('\\n  async ' + name + '\\(([^)]*)\\) \\{').exec(src);
      let i = m.index + m[0].length, depth = 1; const start = i;
      for (; i < src.length && depth > 0; i++) { if (src[i] === '{') depth++; else if (src[i] === '}') depth--; }
      return new AsyncFunction('BatCall', 'CallTypeClassifier', ...argNames, src.slice(start, i - 1));
    };
    if (c.kind === 'measure') {
      const sp = fns.generateSpectrogramWasm.call(self, x, c.sr, c.flow_khz, c.fhigh_khz);
      const nb = sp.freqBins.length, F = sp.powerMatrix.length;
      const db = new Float32Array(F * nb);
      sp.powerMatrix.forEach((row, f) => db.set(row, f * nb));
      emit(c.name, 'db', db); emit(c.name, 'freq', sp.freqBins); emit(c.name, 'shape', f64([F, nb]));
      emit(c.name, 'time', f64(sp.timeFrames)); emit(c.name, 'freqres', f64([sp.freqResolution]));
      const rows = [], rowsDebug = [];
      for (const q of c.slices) {                      // [firstFrame, lastFrame, useZonalMap]
        const a = Math.max(0, q[0]), b = Math.min(F - 1, q[1]);
        const call = new BatCall();
        call.spectrogram = sp.powerMatrix.slice(a, b + 1);
        call.timeFrames = sp.timeFrames.slice(a, b + 2);
        call.startTime_s = sp.timeFrames[a];
        call.endTime_s = sp.timeFrames[Math.min(b + 1, F - 1)];
        call.calculateDuration();
        const zonal = q[2] ? fns.calculateZonalNoiseFloors.call(self, sp.powerMatrix, sp.freqBins, 0, Math.min(5, F - 1)) : null;
        const fresh = () => Object.assign(new BatCall(), call);
        for (const debug of [false, true]) {           // outside and inside the detector's debug mode
          const k = fresh();
          let threw = 0;
          self.debugMode = debug;
          try { realMeasure.call(self, k, c.flow_khz, c.fhigh_khz, sp.freqBins, sp.freqResolution, zonal); } catch (err) { threw = 1; }
          self.debugMode = false;
          (debug ? rowsDebug : rows).push(...callRow(k), k.isDiscarded ? 1 : 0, threw);
        }
      }
      emit(c.name, 'measured', f64(rows));
      emit(c.name, 'measured_debug', f64(rowsDebug));
    } else if (c.kind === 'fullfile' || c.kind === 'fullfile_measured') {
      const processFullFile = scoped('processFullFile', ['fullAudioData', 'sampleRate', 'flowKHz', 'fhighKHz', 'options']);
      const calls = await processFullFile.call(self, BatCall, CallTypeClassifier, x, c.sr, c.flow_khz, c.fhigh_khz,
        { threshold_dB: c.scan_threshold_db, padding_ms: c.padding_ms });
      emit(c.name, 'calls', f64([].concat(...calls.map((k) => [k.startTime_s, k.endTime_s, k.duration_ms, k.snr_dB, k.spectrogram.length]))));
      emit(c.name, 'peaks', f64([].concat(...calls.map((k) => [k.peakFreq_kHz, k.peakFreq_ms, k.peakFrameIdx, k.peakPower_dB]))));
      emit(c.name, 'measured', f64([].concat(...calls.map((k) => callRow(k)))));
      emit(c.name, 'measure_threw', f64([measureThrew]));
      emit(c.name, 'discarded', f64(captured.map((k) => k.discarded)));
      emit(c.name, 'zonal_first', captured.length ? captured[0].zonal : new Float64Array(0));
      emit(c.name, 'measured_frames', f64(captured.map((k) => k.frames)));
    } else if (c.kind === 'calls_measured') {
      // the whole detectCalls: measurement stage, classifier and SNR filter loop as the reference has them
      const detectCalls = scoped('detectCalls', ['audioData', 'sampleRate', 'flowKHz', 'fhighKHz', 'options']);
      let noise = null;
      if (c.noise_len > 0) noise = fns.generateSpectrogramWasm.call(self, inputs.subarray(c.noise_off, c.noise_off + c.noise_len), c.sr, c.flow_khz, c.fhigh_khz);
      const types = ['CF', 'FM', 'CF-FM'], qualities = ['Very Poor', 'Poor', 'Normal', 'Good', 'Excellent'];
      for (const skip of [false, true]) {
        const calls = await detectCalls.call(self, BatCall, CallTypeClassifier, x, c.sr, c.flow_khz, c.fhigh_khz, { skipSNR: skip, noiseSpectrogram: noise });
        emit(c.name, skip ? 'measured_skip_snr' : 'measured', f64([].concat(...calls.map((k) => [...callRow(k), k.noiseFloor_dB,
          k.isDiscarded ? 1 : 0, types.indexOf(k.callType), qualities.indexOf(k.quality), k.Flow, k.Fhigh === null ? NaN : k.Fhigh]))));
      }
    } else if (c.kind === 'calls') {
      const detectCalls = scoped('detectCalls', ['audioData', 'sampleRate', 'flowKHz', 'fhighKHz', 'options']);
      let noise = null;
      if (c.noise_len > 0) noise = fns.generateSpectrogramWasm.call(self, inputs.subarray(c.noise_off, c.noise_off + c.noise_len), c.sr, c.flow_khz, c.fhigh_khz);
      const calls = await detectCalls.call(self, BatCall, CallTypeClassifier, x, c.sr, c.flow_khz, c.fhigh_khz, { skipSNR: true, noiseSpectrogram: noise });
      emit(c.name, 'calls', f64([].concat(...calls.map((k) => [k.startTime_s, k.endTime_s, k.duration_ms, k.noiseFloor_dB, k.snr_dB, k.spectrogram.length]))));
      emit(c.name, 'measured_frames', f64(captured.map((k) => k.frames)));
      emit(c.name, 'peaks', f64([].concat(...calls.map((k) => [k.peakFreq_kHz, k.peakFreq_ms, k.peakFrameIdx, k.peakPower_dB]))));
      emit(c.name, 'zonal', captured.length ? captured[0].zonal : new Float64Array(0));
    } else if (c.kind === 'refine') {
      const out = [];
      for (const [s0, s1] of c.ranges) out.push(fns.refineEndUsingOscillogram.call(self, x, c.sr, s0, s1));
      emit(c.name, 'refined', f64(out));
    } else if (c.kind === 'zonal') {
      const sp = fns.generateSpectrogramWasm.call(self, x, c.sr, c.flow_khz, c.fhigh_khz);
      const nb = sp.freqBins.length, F = sp.powerMatrix.length;
      const db = new Float32Array(F * nb);
      sp.powerMatrix.forEach((row, f) => db.set(row, f * nb));
      emit(c.name, 'db', db);
      emit(c.name, 'freq', sp.freqBins);
      emit(c.name, 'shape', f64([F, nb]));
      for (const [f0, f1] of c.ranges) emit(c.name, 'zonal_' + f0 + '_' + f1, zoneArray(fns.calculateZonalNoiseFloors.call(self, sp.powerMatrix, sp.freqBins, f0, f1)));
      // the noise floor of detectCalls (batCallDetector.js:1223-1235): 25th percentile of every value, floored at -80
      const all = Array.from(db); all.sort((a, b) => a - b);
      emit(c.name, 'p25', f64([all[Math.max(0, Math.floor(all.length * 0.25))]]));
      // per-frame threshold crossings with interpolation
      const scans = [];
      for (const thr of c.thresholds) for (let f = 0; f < F; f += c.scan_stride) for (const dir of ['HighToLow', 'LowToHigh']) {
        const r = fns._scanSpectrumForFrequency.call(self, sp.powerMatrix[f], sp.freqBins, thr, dir);
        scans.push(r ? r.freq_Hz : -1, r ? r.binIdx : -1);
      }
      emit(c.name, 'scans', f64(scans));
    } else if (c.kind === 'thresholds') {
      const sp = fns.generateSpectrogramWasm.call(self, x, c.sr, c.flow_khz, c.fhigh_khz);
      const nb = sp.freqBins.length, F = sp.powerMatrix.length;
      const db = new Float32Array(F * nb);
      sp.powerMatrix.forEach((row, f) => db.set(row, f * nb));
      emit(c.name, 'db', db); emit(c.name, 'freq', sp.freqBins); emit(c.name, 'shape', f64([F, nb]));
      emit(c.name, 'time', f64(sp.timeFrames));
      const num = (v) => (v === null || v === undefined ? NaN : (v === true ? 1 : v === false ? 0 : v));
      const lowPlain = [], highPlain = [], lowDebug = [], highDebug = [];
      for (const debug of [false, true]) for (const q of c.slices) {   // [firstFrame, lastFrame, useZonalMap, limitFrame (-1 = null)]
        const low = debug ? lowDebug : lowPlain, high = debug ? highDebug : highPlain;
        self.debugMode = debug;
        const a = Math.max(0, q[0]), b = Math.min(F - 1, q[1]);
        const spec = sp.powerMatrix.slice(a, b + 1), tf = sp.timeFrames.slice(a, b + 2);
        let peak = -Infinity, peakFrame = 0;
        spec.forEach((row, f) => { for (let k = 0; k < row.length; k++) if (row[k] > peak) { peak = row[k]; peakFrame = f; } });
        const zonal = q[2] ? fns.calculateZonalNoiseFloors.call(self, sp.powerMatrix, sp.freqBins, 0, Math.min(5, F - 1)) : null;
        const limit = q[3] < 0 ? null : q[3];
        let r;
        try {
          r = fns.findOptimalLowFrequencyThreshold.call(self, spec, tf, sp.freqBins, c.flow_khz, c.fhigh_khz, peak, peakFrame, limit, zonal);
          low.push(num(r.threshold), num(r.lowFreq_Hz), num(r.endFreq_Hz), num(r.lowFreqFrameIdx), num(r.lowFreqBinIdx), num(r.warning), num(r.discard), 0);
        } catch (err) { low.push(NaN, NaN, NaN, NaN, NaN, NaN, NaN, 1); }
        try {
          r = fns.findOptimalHighFrequencyThreshold.call(self, spec, tf, sp.freqBins, c.flow_khz, c.fhigh_khz, peak, peakFrame, zonal);
          high.push(num(r.threshold), num(r.highFreq_Hz), num(r.highFreqBinIdx), num(r.highFreqFrameIdx), num(r.startFreq_Hz),
            num(r.finalSearchLimitFrame), num(r.warning), num(r.isCFStablePattern), 0);
        } catch (err) { high.push(NaN, NaN, NaN, NaN, NaN, NaN, NaN, NaN, 1); }
        self.debugMode = false;
      }
      emit(c.name, 'low', f64(lowPlain)); emit(c.name, 'high', f64(highPlain));
      emit(c.name, 'low_debug', f64(lowDebug)); emit(c.name, 'high_debug', f64(highDebug));
    } else if (c.kind === 'snr') {
      const sp = fns.generateSpectrogramWasm.call(self, x, c.sr, c.flow_khz, c.fhigh_khz);
      const nb = sp.freqBins.length, F = sp.powerMatrix.length;
      const db = new Float32Array(F * nb);
      sp.powerMatrix.forEach((row, f) => db.set(row, f * nb));
      emit(c.name, 'db', db); emit(c.name, 'freq', sp.freqBins); emit(c.name, 'shape', f64([F, nb]));
      let noise = null;
      if (c.noise_len > 0) {
        noise = fns.generateSpectrogramWasm.call(self, inputs.subarray(c.noise_off, c.noise_off + c.noise_len), c.sr, c.flow_khz, c.fhigh_khz);
        const nF = noise.powerMatrix.length, ndb = new Float32Array(nF * nb);
        noise.powerMatrix.forEach((row, f) => ndb.set(row, f * nb));
        emit(c.name, 'noise_db', ndb); emit(c.name, 'noise_shape', f64([nF, nb]));
      }
      const num = (v) => (v === null || v === undefined ? NaN : v);
      const rows = [];
      for (const q of c.queries) for (const useNoise of [false, true]) {
        if (useNoise && !noise) continue;
        const r = fns.calculateRMSbasedSNR.call(self, { lowFreq_kHz: q[0], highFreq_kHz: q[1] }, sp.powerMatrix, sp.freqBins, q[2], q[3],
          c.flow_khz, c.fhigh_khz, useNoise ? noise : null);
        rows.push(num(r.snr_dB), num(r.signalPowerMean_dB), num(r.noisePowerMean_dB), r.signalCount, r.noiseCount,
          num(r.debug.signalThreshold), num(r.debug.signalMax), num(r.debug.signalMin), useNoise ? 1 : 0);
      }
      emit(c.name, 'snr', f64(rows));
    } else if (c.kind === 'filter') {
      for (const [hz, order] of c.filters) emit(c.name, 'hp_' + hz + '_' + order, fns.applyHighpassFilter.call(self, x, hz, c.sr, order));
      emit(c.name, 'auto', f64(c.peaks_khz.map((p) => fns.calculateAutoHighpassFilterFreq.call(self, p))));
      const sg = fns.savitzkyGolay.call(self, Array.from(x.subarray(0, c.sg_len)).map((v) => v * 1000));
      emit(c.name, 'sg', f64(sg));
      emit(c.name, 'sg_short', f64(fns.savitzkyGolay.call(self, [1, 2, 4])));
    }
    e.free();
  }
}
run().then(() => {
  fs.closeSync(outFd);
  fs.writeFileSync(path.join(work, 'outputs.json'), JSON.stringify(manifest));
  process.stdout.write('framing_driver: ' + cases.length + ' cases, ' + outPos + ' bytes\n');
}).catch((err) => { process.stderr.write(String(err && err.stack || err) + '\n'); process.exit(1); });
