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

const AsyncFunction = Object.getPrototypeOf(async function () {}).constructor;
// Text of one method of the plugin class, as a function; `scope` names free identifiers of the module it reads.
function method(name, scope) {
  const m = new RegExp('\\n    (async )?' + name + '\\(([^)]*)\\) \\{').exec(src);
  if (!m) throw new Error(name + ' not found');
  let i = m.index + m[0].length, depth = 1;
  const start = i;
  for (; i < src.length && depth > 0; i++) {
    if (src[i] === '{') depth++;
    else if (src[i] === '}') depth--;
  }
  const args = m[2].split(',').map((t) => t.trim()).filter((t) => t);
  const names = Object.keys(scope || {});
  const f = new (m[1] ? AsyncFunction : Function)(...names, ...args, src.slice(start, i - 1));
  const bound = names.map((k) => scope[k]);
  return function (...a) { return f.call(this, ...bound, ...a); };
}
const getFrequencies = method('getFrequencies');
// the scale helpers and the filter-bank builder (spectrogram.esm.js:1197-1273, 1318-1349); `n` is the module's ERB constant (:344)
const erbConst = new Function('return ' + /\nconst n = ([^;]+);/.exec(src)[1])();
const scaleMethods = {};
for (const k of ['createFilterBank', 'hzToMel', 'melToHz', 'hzToLog', 'logToHz', 'hzToBark', 'barkToHz', 'hzToErb', 'erbToHz',
  'flattenAndLoadFilterBank']) scaleMethods[k] = method(k, (k === 'hzToErb' || k === 'erbToHz') ? { n: erbConst } : {});

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
      noverlap: c.noverlap, canvas: { width: c.canvas_width },
      frequencyMax: c.frequency_max || 0, frequencyMin: c.frequency_min || 0,
      _wasmReady: Promise.resolve(), wrapper: {}, scale: c.scale || 'linear', _loadedFilterBankKey: null,
      _lastFilterBankScale: null, _filterBankCacheByKey: {}, _filterBankCache: {}, _filterBankFlat: null, _filterBankMatrix: null,
      numMelFilters: c.fft / 2, numLogFilters: c.fft / 2, numBarkFilters: c.fft / 2, numErbFilters: c.fft / 2,   // :415-418
      gainDB: c.gain_db, rangeDB: c.range_db,
      _wasmEngine: {
        compute_spectrogram_u8: (x, v, g, r) => e.u8(x, v, g, r),
        get_global_max: () => e.globalMax(),
        get_peak_magnitudes: (r) => e.peakMags(r),
        get_num_filters: () => e.numFilters(),
        load_filter_bank: (w, nf) => e.loadFilterBank(w, nf),
        clear_filter_bank: () => e.clearFilterBank(),
      },
    };
    Object.assign(self, scaleMethods);
    const buffer = {
      sampleRate: c.sr, numberOfChannels: chans.length, length: chans[0].length,
      getChannelData: (k) => chans[k],
    };
    const frames = await getFrequencies.call(self, buffer, false);
    result[c.name] = {
      frames: frames.map((ch) => ch.map((f) => Buffer.from(f).toString('base64'))),
      peaks: self.peakBandArrayPerChannel.map((ch) => ch.map((fr) => fr.map((p) => [p.bin, p.magnitude, p.isMainPeak ? 1 : 0]))),
      bank: self._filterBankFlat ? Buffer.from(self._filterBankFlat.buffer).toString('base64') : null,
    };
    e.free();
  }
  fs.writeFileSync(path.join(work, 'outputs.json'), JSON.stringify(result));
  process.stdout.write('plugin_driver: ' + cases.length + ' cases\n');
})().catch((err) => { process.stderr.write(String(err && err.stack || err) + '\n'); process.exit(1); });
