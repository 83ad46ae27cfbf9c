// Test infrastructure only: runs the reference's GUANO reader (/root/reference/modules/guanoReader.js, read as text at
// run time; its two `export` keywords are dropped so that Node 12 can evaluate it as a script) on the files of
// cases.json and writes what it returns.
// usage: node guano_driver.js <guanoReader.js> <workdir>
'use strict';
const fs = require('fs');
const path = require('path');
const src = fs.readFileSync(process.argv[2], 'utf8').replace(/^export /gm, '');
const work = process.argv[3];
const mod = new Function(src + '\nreturn { extractGuanoMetadata, parseGuanoMetadata };')();
const cases = JSON.parse(fs.readFileSync(path.join(work, 'cases.json'), 'utf8'));
(async () => {
  const out = {};
  for (const c of cases) {
    const bytes = Buffer.from(c.file, 'base64');
    const ab = bytes.buffer.slice(bytes.byteOffset, bytes.byteOffset + bytes.length);
    let text = null, threw = null;
    try { text = await mod.extractGuanoMetadata({ arrayBuffer: async () => ab }); } catch (err) { threw = err.name; }
    out[c.name] = { text, threw, parsed: text === null ? null : mod.parseGuanoMetadata(text) };
  }
  for (const t of cases.filter((c) => c.input_text !== undefined)) out[t.name].parsed_text = mod.parseGuanoMetadata(t.input_text);
  fs.writeFileSync(path.join(work, 'outputs.json'), JSON.stringify(out));
  process.stdout.write('guano_driver: ' + cases.length + ' cases\n');
})().catch((err) => { process.stderr.write(String(err && err.stack || err) + '\n'); process.exit(1); });
