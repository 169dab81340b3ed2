// TEST INFRASTRUCTURE — live-reference harness (not product code, never shipped).
//
// A fake "plotly.js" for kaleido's headless Chromium + SwiftShader: kaleido calls
// Plotly.toImage(fig, opts) and returns whatever string the promise resolves to.
// We use that hook to load the UNMODIFIED reference page script
// (/root/reference/script.js) into a synthetic DOM, drive its own globals
// (config, initFramebuffers, splat, multipleSplats, step, framebufferToTexture, the
// *Program objects and blit) and hand back fp32 dumps of its five simulation fields.
//
// fig.layout is the scenario:
//   canvasW, canvasH   CSS px size of the canvas (sets the aspect ratio)
//   config             overrides for the reference's `config` object
//   seed               mulberry32 seed that replaces Math.random
//   randomSplats       n -> multipleSplats(n) after the fields are re-created
//   splats             [[x,y,dx,dy,r,g,b], ...] explicit splat() calls
//   inject             {velocity|pressure|divergence|curl|dye: base64 fp32} exact state upload
//   synth              {velocity|pressure|divergence|curl|dye: {seed, noise, amp: [per channel], cx, cy, R2}} exact state generated
//                      IN THE PAGE (grids too large to upload, 4096^2): texel (i, j), channel c gets
//                      fround(amp[c] * shape_c(x, y) + (mulberry32_k - 0.5) * noise), x = (i + .5) / W, y = (j + .5) / H,
//                      shapes: a compact vortex (-(y - cy) g, (x - cx) g, g, x y with g = max(0, 1 - r2 / R2)^2), k = (j W + i) nch + c;
//                      doubles and + - * / max only, so tests/synth.py reproduces every value bit for bit in numpy
//   passes             ["curl","vorticity",...] run single passes instead of step()
//   resizeTo           {SIM_RESOLUTION, DYE_RESOLUTION} -> initFramebuffers() again after the steps
//   steps, dt, timing, noDump
//   halfTargets        true: every draw into a simulation framebuffer is followed by an fp16 round trip of that attachment
//                      (half-float render targets emulated around the unmodified reference: its shaders, a 16F target's rounding)
//   sample             {stride, band: [row0, row1], bands: [[row0, row1], ...]}: instead of the full dumps, per field every stride-th row / column, the rows of
//                      the band at full resolution and max |value| (grids too large to return whole, e.g. 4096^2)
//   render             {config: {SHADING, BLOOM, SUNRAYS, TRANSPARENT, BACK_COLOR, BLOOM_*, SUNRAYS_*, CAPTURE_RESOLUTION},
//                       dither: {w, h, seed}} -> after the steps: the reference's captureScreenshot() up to the PNG
//                      (render(target) into a float FBO of getResolution(CAPTURE_RESOLUTION), framebufferToTexture,
//                      normalizeTexture); dumps the float frame, the 8-bit frame, and the bloom / sunrays buffers.
//                      `dither` replaces the blue-noise PNG (an asset, not shipped) by a seeded R8 pattern.
//   frames             [{dt, events: [{type, offsetX, offsetY, touches: [{identifier, pageX, pageY}], code, key}]}]:
//                      per frame the events go through the reference's OWN listeners (script.js:1464-1530) and then
//                      the body of its update() runs with that dt (updateColors, applyInputs, step unless PAUSED)
(function () {
  function load(src) {
    return new Promise(function (ok, no) {
      var s = document.createElement('script'); s.src = src; s.onload = ok;
      s.onerror = function () { no(new Error('cannot load ' + src)); };
      document.head.appendChild(s);
    });
  }
  function b64(f32) {
    var u = new Uint8Array(f32.buffer, f32.byteOffset, f32.byteLength), s = '';
    for (var i = 0; i < u.length; i += 32768) s += String.fromCharCode.apply(null, u.subarray(i, i + 32768));
    return btoa(s);
  }
  function unb64(str) {
    var bin = atob(str), u = new Uint8Array(bin.length);
    for (var i = 0; i < bin.length; i++) u[i] = bin.charCodeAt(i);
    return new Float32Array(u.buffer);
  }
  // fp32 -> nearest fp16 (ties to even, subnormals kept, overflow to infinity) -> fp32: what a half-float render target keeps
  var _f = new Float32Array(1), _u = new Uint32Array(_f.buffer);
  function roundHalf(v) {
    _f[0] = v;
    var x = _u[0], sign = x & 0x80000000, h, e, m, shift, r, rem, half;
    x = x & 0x7fffffff;
    if (x > 0x7f800000) return v;                                     // NaN
    if (x >= 0x47800000) { _u[0] = (sign | 0x7f800000) >>> 0; return _f[0]; }   // >= 65536 -> inf
    if (x >= 0x38800000) {                                            // normal half (may round up to inf at 65520)
      x = (x + 0xfff + ((x >>> 13) & 1)) >>> 0;
      x = (x & 0xffffe000) >>> 0;
      _u[0] = (sign | x) >>> 0;
      if ((x >>> 0) >= 0x47800000) _u[0] = (sign | 0x7f800000) >>> 0;
      return _f[0];
    }
    if (x <= 0x33000000) { _u[0] = sign >>> 0; return _f[0]; }        // <= 2^-25 -> +-0
    e = x >>> 23; m = (x & 0x7fffff) | 0x800000; shift = 126 - e;     // subnormal half: unit 2^-24
    r = m >>> shift; rem = m & ((1 << shift) - 1); half = 1 << (shift - 1);
    if (rem > half || (rem === half && (r & 1))) r++;
    h = r * 5.9604644775390625e-08;                                   // r * 2^-24, exact
    return sign ? -h : h;
  }
  function el(tag, cls, id) {
    var e = document.createElement(tag); if (cls) e.className = cls; if (id) e.id = id;
    document.body.appendChild(e); return e;
  }

  window.Plotly = { version: '2.0.0', toImage: function (fig) {
    var P = fig.layout, out = {};
    el('div', 'promo').appendChild(document.createElement('span')).className = 'promo-close';
    el('a', null, 'apple_link'); el('a', null, 'google_link');
    var cv = document.createElement('canvas'); cv.style.display = 'block';
    cv.style.width = (P.canvasW || 512) + 'px'; cv.style.height = (P.canvasH || 512) + 'px';
    document.body.insertBefore(cv, document.body.firstChild);
    window.ga = function () {}; window.requestAnimationFrame = function () { return 0; };
    var seed = 0;
    function reseed() { seed = (P.seed === undefined ? 1234 : P.seed) >>> 0; }
    var draws = [];
    Math.random = function () {
      seed |= 0; seed = seed + 0x6D2B79F5 | 0;
      var t = Math.imul(seed ^ seed >>> 15, 1 | seed);
      t = t + Math.imul(t ^ t >>> 7, 61 | t) ^ t;
      var r = ((t ^ t >>> 14) >>> 0) / 4294967296;
      draws.push(r); return r;
    };
    reseed();
    var refdir = P.refdir || 'file:///root/reference/';
    return load(refdir + 'dat.gui.min.js').then(function () { return load(refdir + 'script.js'); }).then(function () {
      var k;
      for (k in (P.config || {})) config[k] = P.config[k];
      if (P.halfTargets) {
        // Emulate half-float render targets (what halfFloatTexType gives the reference on a real GPU; SwiftShader keeps fp32):
        // after EVERY draw of the reference into one of its simulation framebuffers, the attachment is read back, rounded to
        // fp16 and written back — the reference's own shaders, plus the store rounding of a 16F target.  script.js is untouched:
        // only the context's drawElements / bindFramebuffer methods are wrapped.
        var curFbo = null, rawBind = gl.bindFramebuffer.bind(gl), rawDraw = gl.drawElements.bind(gl);
        gl.bindFramebuffer = function (t, f) { curFbo = f; rawBind(t, f); };
        gl.drawElements = function (a, b, c, d) {
          rawDraw(a, b, c, d);
          if (!curFbo) return;
          var cands = [[dye && dye.read, 4], [dye && dye.write, 4], [velocity && velocity.read, 2], [velocity && velocity.write, 2],
                       [pressure && pressure.read, 1], [pressure && pressure.write, 1], [divergence, 1], [curl, 1]], i, t = null, nch = 0;
          for (i = 0; i < cands.length; i++) if (cands[i][0] && cands[i][0].fbo === curFbo) { t = cands[i][0]; nch = cands[i][1]; }
          if (!t) return;
          var w = t.width, h = t.height, px4 = new Float32Array(w * h * 4), d2 = new Float32Array(w * h * nch), n, c2;
          gl.readPixels(0, 0, w, h, gl.RGBA, gl.FLOAT, px4);
          for (n = 0; n < w * h; n++) for (c2 = 0; c2 < nch; c2++) d2[n * nch + c2] = roundHalf(px4[n * 4 + c2]);
          // leave the reference's texture-unit state exactly as it was (step() attaches the divergence texture ONCE in front of
          // its Jacobi loop, script.js:1261): work on unit 7 and put its binding and the active unit back
          var act = gl.getParameter(gl.ACTIVE_TEXTURE);
          gl.activeTexture(gl.TEXTURE7);
          var was = gl.getParameter(gl.TEXTURE_BINDING_2D);
          gl.bindTexture(gl.TEXTURE_2D, t.texture);
          gl.texSubImage2D(gl.TEXTURE_2D, 0, 0, 0, w, h, nch === 1 ? gl.RED : nch === 2 ? gl.RG : gl.RGBA, gl.FLOAT, d2);
          gl.bindTexture(gl.TEXTURE_2D, was);
          gl.activeTexture(act);
        };
      }
      dye = null; velocity = null; initFramebuffers();
      reseed(); draws.length = 0;

      // record every splat() the reference issues (multipleSplats goes through the global)
      var splatLog = [], refSplat = window.splat;
      window.splat = function (x, y, dx, dy, c) { splatLog.push([x, y, dx, dy, c.r, c.g, c.b]); return refSplat(x, y, dx, dy, c); };

      function upload(target, comps, data) {
        // exact fp32 upload: a harness-owned 32F NEAREST texture drawn through the
        // reference's own copyProgram into the (half-float-declared, fp32-stored) target
        var ifmt = comps === 1 ? gl.R32F : comps === 2 ? gl.RG32F : gl.RGBA32F;
        var fmt = comps === 1 ? gl.RED : comps === 2 ? gl.RG : gl.RGBA;
        gl.activeTexture(gl.TEXTURE0);
        var tex = gl.createTexture(); gl.bindTexture(gl.TEXTURE_2D, tex);
        gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_MIN_FILTER, gl.NEAREST);
        gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_MAG_FILTER, gl.NEAREST);
        gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_WRAP_S, gl.CLAMP_TO_EDGE);
        gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_WRAP_T, gl.CLAMP_TO_EDGE);
        gl.texImage2D(gl.TEXTURE_2D, 0, ifmt, target.width, target.height, 0, fmt, gl.FLOAT, data);
        copyProgram.bind();
        gl.uniform1i(copyProgram.uniforms.uTexture, 0);
        gl.activeTexture(gl.TEXTURE0); gl.bindTexture(gl.TEXTURE_2D, tex);
        blit(target);
        gl.deleteTexture(tex);
      }
      var inj = P.inject || {};
      if (inj.velocity) upload(velocity.read, 2, unb64(inj.velocity));
      if (inj.pressure) upload(pressure.read, 1, unb64(inj.pressure));
      if (inj.divergence) upload(divergence, 1, unb64(inj.divergence));
      if (inj.curl) upload(curl, 1, unb64(inj.curl));
      if (inj.dye) upload(dye.read, 4, unb64(inj.dye));
      function synth(target, nch, spec) {
        var w = target.width, h = target.height, a = new Float32Array(w * h * nch), st = (spec.seed >>> 0), i, j, c, k = 0;
        var amp = spec.amp || [0, 0, 0, 0], cx = spec.cx === undefined ? 0.5 : spec.cx, cy = spec.cy === undefined ? 0.5 : spec.cy;
        var R2 = spec.R2 === undefined ? 0.1 : spec.R2, noise = spec.noise || 0;
        for (j = 0; j < h; j++) {
          var y = (j + 0.5) / h;
          for (i = 0; i < w; i++) {
            var x = (i + 0.5) / w, dx = x - cx, dy = y - cy, r2 = dx * dx + dy * dy, g = Math.max(0, 1 - r2 / R2); g = g * g;
            for (c = 0; c < nch; c++) {
              st = (st + 0x6D2B79F5) | 0;
              var t = Math.imul(st ^ (st >>> 15), 1 | st);
              t = (t + Math.imul(t ^ (t >>> 7), 61 | t)) ^ t;
              var r = ((t ^ (t >>> 14)) >>> 0) / 4294967296;
              var sh = c === 0 ? -dy * g : c === 1 ? dx * g : c === 2 ? g : x * y;
              a[k++] = amp[c] * sh + (r - 0.5) * noise;
            }
          }
        }
        upload(target, nch, a);
      }
      var syn = P.synth || {};
      if (syn.velocity) synth(velocity.read, 2, syn.velocity);
      if (syn.pressure) synth(pressure.read, 1, syn.pressure);
      if (syn.divergence) synth(divergence, 1, syn.divergence);
      if (syn.curl) synth(curl, 1, syn.curl);
      if (syn.dye) synth(dye.read, 4, syn.dye);

      if (P.randomSplats) multipleSplats(P.randomSplats);
      (P.splats || []).forEach(function (s) { splat(s[0], s[1], s[2], s[3], { r: s[4], g: s[5], b: s[6] }); });

      var dt = P.dt === undefined ? 0.016666 : P.dt;
      // single-pass replays: the bind/uniform/blit sequence of the matching block of step()
      var PASS = {
        curl: function () {
          curlProgram.bind();
          gl.uniform2f(curlProgram.uniforms.texelSize, velocity.texelSizeX, velocity.texelSizeY);
          gl.uniform1i(curlProgram.uniforms.uVelocity, velocity.read.attach(0));
          blit(curl);
        },
        vorticity: function () {
          vorticityProgram.bind();
          gl.uniform2f(vorticityProgram.uniforms.texelSize, velocity.texelSizeX, velocity.texelSizeY);
          gl.uniform1i(vorticityProgram.uniforms.uVelocity, velocity.read.attach(0));
          gl.uniform1i(vorticityProgram.uniforms.uCurl, curl.attach(1));
          gl.uniform1f(vorticityProgram.uniforms.curl, config.CURL);
          gl.uniform1f(vorticityProgram.uniforms.dt, dt);
          blit(velocity.write); velocity.swap();
        },
        divergence: function () {
          divergenceProgram.bind();
          gl.uniform2f(divergenceProgram.uniforms.texelSize, velocity.texelSizeX, velocity.texelSizeY);
          gl.uniform1i(divergenceProgram.uniforms.uVelocity, velocity.read.attach(0));
          blit(divergence);
        },
        clear: function () {
          clearProgram.bind();
          gl.uniform1i(clearProgram.uniforms.uTexture, pressure.read.attach(0));
          gl.uniform1f(clearProgram.uniforms.value, config.PRESSURE);
          blit(pressure.write); pressure.swap();
        },
        jacobi: function () {
          pressureProgram.bind();
          gl.uniform2f(pressureProgram.uniforms.texelSize, velocity.texelSizeX, velocity.texelSizeY);
          gl.uniform1i(pressureProgram.uniforms.uDivergence, divergence.attach(0));
          gl.uniform1i(pressureProgram.uniforms.uPressure, pressure.read.attach(1));
          blit(pressure.write); pressure.swap();
        },
        gradsub: function () {
          gradienSubtractProgram.bind();
          gl.uniform2f(gradienSubtractProgram.uniforms.texelSize, velocity.texelSizeX, velocity.texelSizeY);
          gl.uniform1i(gradienSubtractProgram.uniforms.uPressure, pressure.read.attach(0));
          gl.uniform1i(gradienSubtractProgram.uniforms.uVelocity, velocity.read.attach(1));
          blit(velocity.write); velocity.swap();
        },
        advect_velocity: function () {
          advectionProgram.bind();
          gl.uniform2f(advectionProgram.uniforms.texelSize, velocity.texelSizeX, velocity.texelSizeY);
          var id = velocity.read.attach(0);
          gl.uniform1i(advectionProgram.uniforms.uVelocity, id);
          gl.uniform1i(advectionProgram.uniforms.uSource, id);
          gl.uniform1f(advectionProgram.uniforms.dt, dt);
          gl.uniform1f(advectionProgram.uniforms.dissipation, config.VELOCITY_DISSIPATION);
          blit(velocity.write); velocity.swap();
        },
        advect_dye: function () {
          advectionProgram.bind();
          gl.uniform2f(advectionProgram.uniforms.texelSize, velocity.texelSizeX, velocity.texelSizeY);
          gl.uniform1i(advectionProgram.uniforms.uVelocity, velocity.read.attach(0));
          gl.uniform1i(advectionProgram.uniforms.uSource, dye.read.attach(1));
          gl.uniform1f(advectionProgram.uniforms.dt, dt);
          gl.uniform1f(advectionProgram.uniforms.dissipation, config.DENSITY_DISSIPATION);
          blit(dye.write); dye.swap();
        }
      };

      var px = new Float32Array(4), ms = [];
      function sync() {
        [velocity.read, dye.read].forEach(function (t) {
          gl.bindFramebuffer(gl.FRAMEBUFFER, t.fbo); gl.readPixels(0, 0, 1, 1, gl.RGBA, gl.FLOAT, px);
        });
      }
      sync();
      gl.disable(gl.BLEND);
      (P.passes || []).forEach(function (name) { PASS[name](); });
      for (var i = 0; i < (P.steps || 0); i++) {
        var t0 = performance.now(); step(dt);
        if (P.timing) { sync(); ms.push(performance.now() - t0); }
      }
      // input replay: synthetic events carry exactly the properties the reference's handlers read
      function fire(ev) {
        var e = new Event(ev.type, { bubbles: true, cancelable: true });
        ['offsetX', 'offsetY', 'code', 'key'].forEach(function (n) { if (ev[n] !== undefined) Object.defineProperty(e, n, { value: ev[n] }); });
        if (ev.touches) {
          Object.defineProperty(e, 'targetTouches', { value: ev.touches });
          Object.defineProperty(e, 'changedTouches', { value: ev.touches });
        }
        var onWindow = ev.type === 'mouseup' || ev.type === 'touchend' || ev.type === 'keydown';
        (onWindow ? window : canvas).dispatchEvent(e);
      }
      if (P.frames) {
        colorUpdateTimer = 0.0;
        pointers.length = 0; pointers.push(new pointerPrototype());
        splatStack.length = 0;
        out.frameLog = [];
        P.frames.forEach(function (f) {
          (f.events || []).forEach(fire);
          var n0 = splatLog.length;
          updateColors(f.dt); applyInputs();                 // update(), script.js:1176-1186, with a prescribed dt
          if (!config.PAUSED) step(f.dt);
          out.frameLog.push({ splats: splatLog.length - n0, paused: !!config.PAUSED, pointers: pointers.length, draws: draws.length });
        });
      }
      if (P.resizeTo) {
        for (k in P.resizeTo) config[k] = P.resizeTo[k];
        initFramebuffers();
      }
      if (P.render) {
        var R = P.render;
        for (k in (R.config || {})) config[k] = R.config[k];
        updateKeywords();
        initBloomFramebuffers(); initSunraysFramebuffers();
        if (R.dither) {
          var dw = R.dither.w, dh = R.dither.h, ds = (R.dither.seed >>> 0), bytes = new Uint8Array(dw * dh * 3);
          for (var q = 0; q < dw * dh; q++) {           // mulberry32 -> one byte per texel, replicated to RGB
            ds |= 0; ds = ds + 0x6D2B79F5 | 0;
            var tt = Math.imul(ds ^ ds >>> 15, 1 | ds); tt = tt + Math.imul(tt ^ tt >>> 7, 61 | tt) ^ tt;
            var by = ((tt ^ tt >>> 14) >>> 0) >>> 24;
            bytes[3 * q] = bytes[3 * q + 1] = bytes[3 * q + 2] = by;
          }
          var dtex = gl.createTexture();
          gl.bindTexture(gl.TEXTURE_2D, dtex);
          gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_MIN_FILTER, gl.LINEAR);
          gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_MAG_FILTER, gl.LINEAR);
          gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_WRAP_S, gl.REPEAT);
          gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_WRAP_T, gl.REPEAT);
          gl.pixelStorei(gl.UNPACK_ALIGNMENT, 1);
          gl.texImage2D(gl.TEXTURE_2D, 0, gl.RGB, dw, dh, 0, gl.RGB, gl.UNSIGNED_BYTE, bytes);
          ditheringTexture = { texture: dtex, width: dw, height: dh,
            attach: function (id) { gl.activeTexture(gl.TEXTURE0 + id); gl.bindTexture(gl.TEXTURE_2D, dtex); return id; } };
        }
        var cres = getResolution(config.CAPTURE_RESOLUTION);
        var target = createFBO(cres.width, cres.height, ext.formatRGBA.internalFormat, ext.formatRGBA.format, ext.halfFloatTexType, gl.NEAREST);
        render(target);
        var ftex = framebufferToTexture(target);
        out.frame = b64(ftex); out.frameSize = [target.width, target.height];
        var n8 = normalizeTexture(ftex, target.width, target.height), s8 = '';
        for (var z = 0; z < n8.length; z += 32768) s8 += String.fromCharCode.apply(null, n8.subarray(z, z + 32768));
        out.frame8 = btoa(s8);
        out.bloom = b64(framebufferToTexture(bloom)); out.bloomSize = [bloom.width, bloom.height];
        out.bloomLevels = bloomFramebuffers.map(function (f) { return [f.width, f.height]; });
        out.sunrays = b64(framebufferToTexture(sunrays)); out.sunraysSize = [sunrays.width, sunrays.height];
        out.mask = b64(framebufferToTexture(dye.write));
        gl.disable(gl.BLEND);
      }
      if (P.probeCoords) {   // what the rasteriser hands the fragment shaders: the varyings of the reference's own baseVertexShader, per texel
        var probe = function (expr) {
          var fs = compileShader(gl.FRAGMENT_SHADER, 'precision highp float; precision highp sampler2D;\n' +
            'varying highp vec2 vUv; varying highp vec2 vL; varying highp vec2 vR; varying highp vec2 vT; varying highp vec2 vB;\n' +
            'void main () { gl_FragColor = ' + expr + '; }');
          var prog = new Program(baseVertexShader, fs);
          prog.bind();
          gl.uniform2f(prog.uniforms.texelSize, velocity.texelSizeX, velocity.texelSizeY);
          blit(dye.write);
          return b64(framebufferToTexture(dye.write));
        };
        out.coords = { a: probe('vec4(vUv.x, vUv.y, vL.x, vR.x)'), b: probe('vec4(vT.y, vB.y, vL.y, vT.x)') };
      }
      out.ms = ms; out.sim = [velocity.width, velocity.height]; out.dye = [dye.width, dye.height];
      out.splats = splatLog; out.draws = draws.length;
      out.canvas = [canvas.width, canvas.height];
      out.gl = { version: gl.getParameter(gl.VERSION), cores: navigator.hardwareConcurrency,
                 maxTex: gl.getParameter(gl.MAX_TEXTURE_SIZE), linear: !!ext.supportLinearFiltering,
                 userAgent: navigator.userAgent };
      var dbg = gl.getExtension('WEBGL_debug_renderer_info');
      if (dbg) out.gl.renderer = gl.getParameter(dbg.UNMASKED_RENDERER_WEBGL);
      if (P.sample) {   // grids too large to ship whole: every `stride`-th row and column, rows [band[0], band[1]) in full, max |value|
        var sampleOf = function (target, nch) {
          var a = framebufferToTexture(target), w = target.width, h = target.height, S = P.sample.stride;
          var bands = P.sample.bands || [P.sample.band], parts = [], nb;
          for (nb = 0; nb < bands.length; nb++) parts.push(a.subarray(bands[nb][0] * w * 4, bands[nb][1] * w * 4));
          var cat = new Float32Array(parts.reduce(function (n, q) { return n + q.length; }, 0)), off = 0;
          for (nb = 0; nb < parts.length; nb++) { cat.set(parts[nb], off); off += parts[nb].length; }
          var sw = Math.ceil(w / S), sh = Math.ceil(h / S), sub = new Float32Array(sw * sh * 4), amax = 0, k, i, j;
          for (j = 0; j < sh; j++) for (i = 0; i < sw; i++) for (k = 0; k < 4; k++) sub[(j * sw + i) * 4 + k] = a[((j * S) * w + i * S) * 4 + k];
          for (i = 0; i < w * h; i++) for (k = 0; k < nch; k++) { var v = Math.abs(a[i * 4 + k]); if (v > amax) amax = v; }
          return { sub: b64(sub), size: [sw, sh], band: b64(cat), absmax: amax };
        };
        out.samples = { velocity: sampleOf(velocity.read, 2), pressure: sampleOf(pressure.read, 1), divergence: sampleOf(divergence, 1),
                        curl: sampleOf(curl, 1), dye: sampleOf(dye.read, 4) };
      }
      if (!P.noDump && !P.sample) out.fields = {
        velocity: b64(framebufferToTexture(velocity.read)), pressure: b64(framebufferToTexture(pressure.read)),
        divergence: b64(framebufferToTexture(divergence)), curl: b64(framebufferToTexture(curl)),
        dye: b64(framebufferToTexture(dye.read)) };
      out.glError = gl.getError();
      return JSON.stringify(out);
    }).catch(function (e) { return JSON.stringify({ error: '' + e + '\n' + (e.stack || '') }); });
  } };
})();
