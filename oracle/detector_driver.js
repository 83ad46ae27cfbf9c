// Test infrastructure only: captures golden vectors of the detector front end
// from the reference's own JavaScript.  The four array passes are methods of
// BatCallDetector in /root/reference/modules/batCallDetector.js; the module as a
// whole cannot be loaded under the Node 12 of the build container (it imports
// DOM modules and uses optional chaining elsewhere), so this script reads the
// file as text at run time, cuts the four method bodies out by name, wraps each
// in a Function and runs it against the reference wasm engine (wasm_ref.js).
// Nothing of the reference is stored in this repository: only inputs and outputs.
// usage: node --experimental-wasm-mv --experimental-wasm-anyref detector_driver.js <wasm> <detector.js> <workdir>
'use strict';
const fs = require('fs');
const path = require('path');
const { load } = require('./wasm_ref.js');

const wasmPath = process.argv[2], jsPath = process.argv[3], work = process.argv[4];
const ref = load(wasmPath);
const src = fs.readFileSync(jsPath, 'utf8');

// "  name(args) {" at class-member indentation -> {args, body}
function method(name) {
  const re = new RegExp('\\n  ' + name + '\\(([^)]*)\\) \\{');
  const m = re.exec(src);
  if (!m) throw new Error('method not found: ' + name);
  let i = m.index + m[0].length, depth = 1;
  const start = i;
  for (; i < src.length && depth > 0; i++) {
    if (src[i] === '{') depth++;
    else if (src[i] === '}') depth--;
  }
  return new Function(...m[1].split(',').map((s) => s.trim().split('=')[0].trim()), src.slice(start, i - 1));
}
const fns = {};
for (const n of ['fastScanSegmentsWasm', 'mergeAndPadSegments', 'generateSpectrogramWasm', 'detectCallSegments']) fns[n] = method(n);

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
const flat = (segs, a, b) => Float64Array.from([].concat(...segs.map((s) => [s[a], s[b]])));

for (const c of cases) {
  const x = inputs.subarray(c.x_off, c.x_off + c.x_len);
  const e = new ref.Engine(c.fft, c.window, null);
  const self = {
    wasmEngine: {
      get_fft_size: () => c.fft,
      get_freq_bins: () => c.fft / 2,
      compute_spectrogram: (a, v) => e.mag(a, v),
    },
    config: { hopPercent: c.hop_percent, callThreshold_dB: c.call_threshold_db, fftSize: c.fft },
  };
  const quiet = console.warn; console.warn = () => {}; console.log = () => {};
  const segs = fns.fastScanSegmentsWasm.call(self, x, c.sr, c.flow_khz, c.fhigh_khz, c.scan_threshold_db);
  emit(c.name, 'scan', flat(segs, 'start', 'end'));
  const merged = fns.mergeAndPadSegments.call(self, segs.map((s) => ({ start: s.start, end: s.end })), x.length, c.sr, c.padding_ms);
  emit(c.name, 'merged', flat(merged, 'startSample', 'endSample'));
  const sp = fns.generateSpectrogramWasm.call(self, x, c.sr, c.flow_khz, c.fhigh_khz);
  console.warn = quiet;
  if (sp === null) {
    emit(c.name, 'db', new Float32Array(0));
    emit(c.name, 'shape', new Float64Array([0, 0]));
  } else {
    const nb = sp.freqBins.length, F = sp.powerMatrix.length;
    const db = new Float32Array(F * nb);
    sp.powerMatrix.forEach((row, f) => db.set(row, f * nb));
    emit(c.name, 'db', db);
    emit(c.name, 'shape', new Float64Array([F, nb]));
    emit(c.name, 'time', Float64Array.from(sp.timeFrames));
    emit(c.name, 'freq', sp.freqBins);
    emit(c.name, 'freqres', new Float64Array([sp.freqResolution]));
    const calls = fns.detectCallSegments.call(self, sp.powerMatrix, sp.timeFrames, sp.freqBins, c.flow_khz, c.fhigh_khz);
    emit(c.name, 'calls', flat(calls, 'startFrame', 'endFrame'));
  }
  e.free();
}
fs.closeSync(outFd);
fs.writeFileSync(path.join(work, 'outputs.json'), JSON.stringify(manifest));
process.stdout.write('detector_driver: ' + cases.length + ' cases, ' + outPos + ' bytes\n');
