// Test infrastructure only: captures golden vectors of the spectrogram plugin's
// frame contract (getFrequencies, /root/reference/modules/spectrogram.esm.js:1364-1529)
// from the reference's own JavaScript.  The plugin module needs a DOM and cannot
// be imported under Node; this script reads the file as text at run time, cuts
// the one method out, and runs it with a stand-in `this` that carries only the
// fields the method reads, over the reference wasm engine (wasm_ref.js).
// usage: node --experimental-wasm-mv --experimental-wasm-anyref plugin_driver.js <wasm> <spectrogram.esm.js> <workdir>
'use strict';
const fs = require('fs');
const path = require('path');
const { load } = require('./wasm_ref.js');

const wasmPath = process.argv[2], jsPath = process.argv[3], work = process.argv[4];
const ref = load(wasmPath);
const src = fs.readFileSync(jsPath, 'utf8');

const m = /\n    async getFrequencies\(([^)]*)\) \{/.exec(src);
if (!m) throw new Error('getFrequencies not found');
let i = m.index + m[0].length, depth = 1;
const start = i;
for (; i < src.length && depth > 0; i++) {
  if (src[i] === '{') depth++;
  else if (src[i] === '}') depth--;
}
const AsyncFunction = Object.getPrototypeOf(async function () {}).constructor;
const getFrequencies = new AsyncFunction('t', 'isRetry', src.slice(start, i - 1));

const cases = JSON.parse(fs.readFileSync(path.join(work, 'cases.json'), 'utf8'));
const inBuf = fs.readFileSync(path.join(work, 'inputs.bin'));
const inputs = new Float32Array(inBuf.buffer, inBuf.byteOffset, inBuf.length / 4);
const result = {};

(async () => {
  console.log = () => {};
  for (const c of cases) {
    const chans = c.channels.map((ch) => inputs.subarray(ch[0], ch[0] + ch[1]));
    const e = new ref.Engine(c.fft, c.window, null);
    const self = {
      options: { peakMode: c.peak_mode, peakThreshold: c.peak_threshold, splitChannels: c.split_channels },
      isDestroyed: false, peakBandArrayPerChannel: [], fftSamples: c.fft, wavesurfer: null,
      frequencyMax: 0, frequencyMin: 0, noverlap: c.noverlap, canvas: { width: c.canvas_width },
      _wasmReady: Promise.resolve(), wrapper: {}, scale: 'linear', _loadedFilterBankKey: null,
      gainDB: c.gain_db, rangeDB: c.range_db,
      _wasmEngine: {
        compute_spectrogram_u8: (x, v, g, r) => e.u8(x, v, g, r),
        get_global_max: () => e.globalMax(),
        get_peak_magnitudes: (r) => e.peakMags(r),
        get_num_filters: () => 0,
      },
    };
    const buffer = {
      sampleRate: c.sr, numberOfChannels: chans.length, length: chans[0].length,
      getChannelData: (k) => chans[k],
    };
    const frames = await getFrequencies.call(self, buffer, false);
    result[c.name] = {
      frames: frames.map((ch) => ch.map((f) => Buffer.from(f).toString('base64'))),
      peaks: self.peakBandArrayPerChannel.map((ch) => ch.map((fr) => fr.map((p) => [p.bin, p.magnitude, p.isMainPeak ? 1 : 0]))),
    };
    e.free();
  }
  fs.writeFileSync(path.join(work, 'outputs.json'), JSON.stringify(result));
  process.stdout.write('plugin_driver: ' + cases.length + ' cases\n');
})().catch((err) => { process.stderr.write(String(err && err.stack || err) + '\n'); process.exit(1); });
