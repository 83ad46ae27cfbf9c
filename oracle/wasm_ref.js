// Test infrastructure only: a minimal loader for the reference's wasm module
// (/root/reference/modules/spectrogram_wasm_bg.wasm), used in the build
// container to capture golden vectors. It talks to the raw wasm exports
// (pointer/length pairs, see modules/spectrogram_wasm_bg.wasm.d.ts:4-38) and
// does not use the reference's generated JS glue.
// Needs: node --experimental-wasm-mv --experimental-wasm-anyref (Node 12-14).
'use strict';
const fs = require('fs');

const NONE_F32 = 0x100000001; // wasm-bindgen encoding of Option<f32>::None (spectrogram_wasm.js:364)

function load(wasmPath) {
  const bytes = fs.readFileSync(wasmPath);
  let ex = null;
  const td = new TextDecoder('utf-8');
  const imports = { wbg: {} };
  const mod = new WebAssembly.Module(bytes);
  for (const imp of WebAssembly.Module.imports(mod)) {
    if (imp.name.indexOf('__wbindgen_throw') >= 0 || imp.name.indexOf('wbindgen_throw') >= 0) {
      imports[imp.module][imp.name] = (p, n) => {
        throw new Error(td.decode(new Uint8Array(ex.memory.buffer, p, n)));
      };
    } else if (imp.name === '__wbindgen_init_externref_table') {
      imports[imp.module][imp.name] = () => {
        const t = ex.__wbindgen_externrefs;
        const off = t.grow(4);
        t.set(0, undefined); t.set(off, undefined); t.set(off + 1, null);
        t.set(off + 2, true); t.set(off + 3, false);
      };
    } else {
      throw new Error('unexpected wasm import ' + imp.module + '.' + imp.name);
    }
  }
  const inst = new WebAssembly.Instance(mod, imports);
  ex = inst.exports;
  ex.__wbindgen_start();

  const putBytes = (u8) => {
    const p = ex.__wbindgen_malloc(u8.length, 1) >>> 0;
    new Uint8Array(ex.memory.buffer).set(u8, p);
    return [p, u8.length];
  };
  const putStr = (s) => putBytes(new TextEncoder().encode(s));
  const putF32 = (a) => {
    const p = ex.__wbindgen_malloc(a.length * 4, 4) >>> 0;
    new Float32Array(ex.memory.buffer).set(a, p / 4);
    return [p, a.length];
  };
  const takeF32 = (r) => {
    const p = r[0] >>> 0, n = r[1] >>> 0;
    const out = new Float32Array(ex.memory.buffer, p, n).slice();
    ex.__wbindgen_free(p, n * 4, 4);
    return out;
  };
  const takeU8 = (r) => {
    const p = r[0] >>> 0, n = r[1] >>> 0;
    const out = new Uint8Array(ex.memory.buffer, p, n).slice();
    ex.__wbindgen_free(p, n, 1);
    return out;
  };
  const takeU16 = (r) => {
    const p = r[0] >>> 0, n = r[1] >>> 0;
    const out = new Uint16Array(ex.memory.buffer, p, n).slice();
    ex.__wbindgen_free(p, n * 2, 2);
    return out;
  };
  const opt = (v) => (v === null || v === undefined) ? NONE_F32 : Math.fround(v);

  class Engine {
    constructor(fftSize, windowName, alpha) {
      const [p, n] = putStr(windowName);
      this.h = ex.spectrogramengine_new(fftSize, p, n, opt(alpha)) >>> 0;
    }
    free() { ex.__wbg_spectrogramengine_free(this.h, 0); this.h = 0; }
    window() { return takeF32(ex.spectrogramengine_get_window_values(this.h)); }
    mag(x, noverlap) { const [p, n] = putF32(x); return takeF32(ex.spectrogramengine_compute_spectrogram(this.h, p, n, noverlap)); }
    u8(x, noverlap, gain, range) { const [p, n] = putF32(x); return takeU8(ex.spectrogramengine_compute_spectrogram_u8(this.h, p, n, noverlap, gain, range)); }
    peaks(r) { return takeU16(ex.spectrogramengine_get_peaks(this.h, r)); }
    peakMags(r) { return takeF32(ex.spectrogramengine_get_peak_magnitudes(this.h, r)); }
    globalMax() { return ex.spectrogramengine_get_global_max(this.h); }
    loadFilterBank(w, nf) { const [p, n] = putF32(w); ex.spectrogramengine_load_filter_bank(this.h, p, n, nf); }
    clearFilterBank() { ex.spectrogramengine_clear_filter_bank(this.h); }
    numFilters() { return ex.spectrogramengine_get_num_filters(this.h) >>> 0; }
  }
  return {
    Engine,
    powerSpectrum(x, sr, fftSize, windowName, overlap) {
      const [p, n] = putF32(x); const [sp, sn] = putStr(windowName);
      return takeF32(ex.compute_power_spectrum(p, n, sr, fftSize, sp, sn, opt(overlap)));
    },
    peakFrequency(spec, sr, fftSize, flow, fhigh) {
      const [p, n] = putF32(spec);
      return ex.find_peak_frequency_from_spectrum(p, n, sr, fftSize, flow, fhigh);
    },
    wavePeaks(x, numPeaks) { const [p, n] = putF32(x); return takeF32(ex.compute_wave_peaks(p, n, numPeaks)); },
    findGlobalMax(x) { const [p, n] = putF32(x); return ex.find_global_max(p, n); },
  };
}
module.exports = { load };
