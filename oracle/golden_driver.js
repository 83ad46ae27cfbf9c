// Test infrastructure only: runs the cases written by make_golden.py through
// the reference wasm (via wasm_ref.js) and dumps every result as raw binary.
// usage: node --experimental-wasm-mv --experimental-wasm-anyref golden_driver.js <wasm> <workdir>
'use strict';
const fs = require('fs');
const path = require('path');
const { load } = require('./wasm_ref.js');

const wasmPath = process.argv[2], work = process.argv[3];
const ref = load(wasmPath);
const cases = JSON.parse(fs.readFileSync(path.join(work, 'cases.json'), 'utf8'));
const inBuf = fs.readFileSync(path.join(work, 'inputs.bin'));
const inputs = new Float32Array(inBuf.buffer, inBuf.byteOffset, inBuf.length / 4);
const outFd = fs.openSync(path.join(work, 'outputs.bin'), 'w');
let outPos = 0;
const manifest = [];

function emit(caseName, key, typed) {
  const b = Buffer.from(typed.buffer, typed.byteOffset, typed.byteLength);
  fs.writeSync(outFd, b);
  const dtype = typed instanceof Float32Array ? 'f32' : typed instanceof Uint16Array ? 'u16' : 'u8';
  manifest.push({ case: caseName, key, dtype, offset: outPos, count: typed.length });
  outPos += b.length;
}
const input = (c) => inputs.subarray(c.x_off, c.x_off + c.x_len);

for (const c of cases) {
  if (c.kind === 'window') {
    const e = new ref.Engine(c.fft, c.window, c.alpha);
    emit(c.name, 'window', e.window());
    e.free();
  } else if (c.kind === 'stft') {
    const e = new ref.Engine(c.fft, c.window, c.alpha);
    const x = input(c);
    if (c.bank_len) e.loadFilterBank(inputs.subarray(c.bank_off, c.bank_off + c.bank_len), c.num_filters);
    emit(c.name, 'mag', e.mag(x, c.noverlap));
    emit(c.name, 'u8', e.u8(x, c.noverlap, c.gain, c.range));
    emit(c.name, 'gmax', new Float32Array([e.globalMax()]));
    c.ratios.forEach((r, i) => {          // key by position: JS prints 0.0 as "0"
      emit(c.name, 'peaks_r' + i, e.peaks(r));
      emit(c.name, 'peakmags_r' + i, e.peakMags(r));
    });
    e.free();
  } else if (c.kind === 'power') {
    const spec = ref.powerSpectrum(input(c), c.sr, c.fft, c.window, c.overlap);
    emit(c.name, 'spectrum', spec);
    const pf = c.queries.map((q) => ref.peakFrequency(spec, c.sr, c.fft, q[0], q[1]));
    emit(c.name, 'peakfreq', new Float32Array(pf));
  } else if (c.kind === 'wave') {
    emit(c.name, 'peaks', ref.wavePeaks(input(c), c.num_peaks));
    emit(c.name, 'gmax', new Float32Array([ref.findGlobalMax(input(c))]));
  } else {
    throw new Error('unknown case kind ' + c.kind);
  }
}
fs.closeSync(outFd);
fs.writeFileSync(path.join(work, 'outputs.json'), JSON.stringify(manifest));
console.log('golden_driver: ' + cases.length + ' cases, ' + outPos + ' bytes');
